for M in bf16 f16 split16; do for K in 0 1 0 1; do
echo "mode $M k2f16 $K: $(NM355_WGRAD_K2F16=$K python bench.py --workload train --conv-mode $M --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
