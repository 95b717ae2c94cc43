cd $GRAFT_REPO_ROOT
NM355_VRNN_MID=0 python tools/time_rollout.py /tmp/r0.pt 2>&1 | grep "MID\|bit"
NM355_VRNN_MID=1 python tools/time_rollout.py /tmp/r1.pt /tmp/r0.pt 2>&1 | grep "MID\|bit"
timeout 900 python -m pytest tests/test_network_gpu.py -x -q -k "rollout or generation or g4 or submodule" 2>&1 | tail -2
