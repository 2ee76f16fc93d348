#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "switches or grouping" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/r04_d_switches.txt
python -m pytest tests/test_gpu_scale.py -x -q -k "index_keyed" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_d_scale.txt
python tools/longread_insert_ab.py 1500000 RB_GROUP_IDX=0 RB_GROUP_IDX=1 > $O/r04_d_longread_ab.txt 2>&1
RB_GROUP_IDX=1 python bench.py --no-cpu-baseline > $O/r04_d_bench_idx1.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/r04_d_bench_idx0.json 2>/dev/null
