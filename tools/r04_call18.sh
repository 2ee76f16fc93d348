#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15 > $O/r04_s_tests.txt
timeout 900 python tools/longread_insert_ab.py 1500000 RB_EARLY_AUTO=0 RB_SWEEP=0 > $O/r04_s_longread_ab.txt 2>&1
