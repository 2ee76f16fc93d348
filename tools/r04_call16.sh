#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "swept or switches" 2>&1 | tail -15 > $O/r04_q_tests.txt
timeout 900 python tools/longread_insert_ab.py 1500000 RB_SWEEP=0 > $O/r04_q_longread_ab.txt 2>&1
