cd $GRAFT_REPO_ROOT
NM355_UP2C_SHAPE=16 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_upsample_composite" 2>&1 | tail -3
NM355_UP2C=0 python tools/time_up2c.py 7
for sh in 32 16; do for d in 0 4; do NM355_UP2C_SHAPE=$sh NM355_UP2C_DIAG=$d python tools/time_up2c.py 7; done; done
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "beyond_fp16" 2>&1 | tail -3
