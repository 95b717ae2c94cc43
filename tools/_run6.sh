cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_upsample" -s 2>&1 | tail -9
NM355_UP2C=0 python tools/time_up2c.py 7
for d in 0 4; do NM355_UP2C_DIAG=$d python tools/time_up2c.py 7; done
