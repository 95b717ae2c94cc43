cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_grad_ops_gpu.py tests/test_train_detector_gpu.py -x -q 2>&1 | tail -3
python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras | cut -c1-200
