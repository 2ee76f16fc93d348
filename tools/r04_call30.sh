#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
RB_SWEEP=1 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/r04_ae_tests_sweep_forced.txt
RB_PF_SKIP=2 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -20 > $O/r04_ae_tests_pfskip_forced.txt
