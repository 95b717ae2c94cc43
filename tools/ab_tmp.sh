python -m pytest tests/test_ops_gpu.py tests/test_grad_ops_gpu.py tests/test_storage16_gpu.py -m gpu -x -q 2>&1 | tail -5
for M in bf16 split16; do for K in 0 1 0 1; do
echo "mode $M convt_f16 $K: $(NM355_CONVT_F16=$K python bench.py --workload train --conv-mode $M --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
for K in 0 1 0 1; do
echo "forward convt_f16 $K: $(NM355_CONVT_F16=$K python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
