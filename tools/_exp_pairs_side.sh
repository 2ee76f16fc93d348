cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sharded_walks.py -x -q 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_multiproc.py tests/test_golden_stage1.py tests/test_gpu_parity.py tests/test_gpu_api_holes.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
