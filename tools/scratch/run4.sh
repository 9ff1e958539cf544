cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_layers.py tests/test_extras.py tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15
