#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_multiproc.py tests/test_gpu_config3.py tests/test_gpu_sharded_walks.py tests/test_gpu_scale.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_n_tests.txt
python tools/loopback_bench.py --ranks 8 --pairs 50000000 > $O/r04_n_loop8_py.txt 2>&1
python tools/loopback_bench.py --ranks 8 --pairs 50000000 --native > $O/r04_n_loop8_native.txt 2>&1
python tools/loopback_bench.py --ranks 8 --pairs 50000000 --trace > $O/r04_n_loop8_py_trace.txt 2>&1
