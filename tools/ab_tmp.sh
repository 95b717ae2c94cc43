python tools/diag_additivity.py f16 2>&1 | tail -9
python -m pytest tests/test_train_detector_gpu.py -m gpu -x -q -k "f16_mode_at_64cubed or seed_103" 2>&1 | tail -3
for K in 0 1 0 1; do
echo "mode bf16 convt_f16 $K: $(NM355_CONVT_F16=$K python bench.py --workload train --conv-mode bf16 --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
