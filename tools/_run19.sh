cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_detector_gpu.py -x -q -s -k "config3 or trajectory or other_keypoint" 2>&1 | grep -v "^$" | tail -8 | cut -c1-300
