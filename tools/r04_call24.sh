#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > $O/r04_y_tests.txt
timeout 900 python bench.py --no-cpu-baseline > $O/r04_y_bench.json 2>/dev/null
timeout 900 python tools/longread_insert_ab.py 1500000 RB_SWEEP=0 > $O/r04_y_longread_ab.txt 2>&1
timeout 1500 python tools/longread_full.py 5000000 > $O/r04_y_longread_full.txt 2>&1
