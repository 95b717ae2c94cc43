cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_upsample" -s > gpurun_out/r2e/ops.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/ops.log
tail -4 gpurun_out/r2e/ops.log
for d in 0 4 5 6 7 12; do
NM355_UP2C_DIAG=$d timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/b$d.log 2>&1
echo "diag=$d $(tail -1 gpurun_out/r2e/b$d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])")"
done
