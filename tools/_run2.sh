cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_upsample" -s > gpurun_out/r2b/ops.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/ops.log
tail -25 gpurun_out/r2b/ops.log
NM355_UP2C=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench_off.log 2>&1
NM355_UP2C=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench_on.log 2>&1
tail -1 gpurun_out/r2b/bench_off.log | cut -c1-900; tail -1 gpurun_out/r2b/bench_on.log | cut -c1-900
