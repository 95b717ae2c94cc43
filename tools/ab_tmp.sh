for K in 0 1 0 1; do
echo "forward clip_late $K: $(NM355_CLIP_LATE=$K python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kypt_l2_vs_cpu"))')"
done
for M in bf16 split16; do for K in 0 1 0 1; do
echo "mode $M clip_late $K: $(NM355_CLIP_LATE=$K python bench.py --workload train --conv-mode $M --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
python -m pytest tests/test_network_gpu.py -m gpu -x -q -k "g1 or g2 or g4 or g5 or config2 or determin" 2>&1 | tail -4
