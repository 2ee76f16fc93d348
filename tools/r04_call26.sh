#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_multiproc.py tests/test_gpu_sharded_walks.py tests/test_gpu_scale.py tests/test_gpu_configs_at_size.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_aa_tests.txt
timeout 600 python tools/loopback_bench.py --ranks 8 --pairs 50000000 > $O/r04_aa_loop8_py.txt 2>&1
timeout 600 python tools/loopback_bench.py --ranks 8 --pairs 50000000 --native > $O/r04_aa_loop8_native.txt 2>&1
timeout 600 python tools/loopback_bench.py --ranks 1 --pairs 50000000 --native > $O/r04_aa_loop1_native.txt 2>&1
