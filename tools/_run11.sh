cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/r2i/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2i/tests.log
tail -5 gpurun_out/r2i/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2i/bench.log 2>&1; tail -1 gpurun_out/r2i/bench.log
