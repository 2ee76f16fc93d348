#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_config3.py tests/test_gpu_sharded_multiproc.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_i_sharded.txt
