cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -2
NM355_UP2C=0 python tools/time_up2c.py 7
for d in 0 4; do NM355_UP2C_DIAG=$d python tools/time_up2c.py 7; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline | cut -c1-200
