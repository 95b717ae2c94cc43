cd $GRAFT_REPO_ROOT
NM355_VRNN_GEMM=0 python tools/time_interp.py 2>&1 | tail -1
NM355_VRNN_GEMM=1 python tools/time_interp.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_network_gpu.py -x -q -s -k "interpolation or generation or submodule or g4 or rollout" 2>&1 | grep -i "passed\|failed\|interpolation\|error" | cut -c1-250
