cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2h/tests.log
tail -4 gpurun_out/r2h/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2h/bench.log 2>&1; tail -1 gpurun_out/r2h/bench.log | cut -c1-250
