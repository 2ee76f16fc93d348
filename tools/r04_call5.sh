#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "switches or grouping" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/r04_e_switches.txt
python -m pytest tests/test_gpu_scale.py tests/test_gpu_bench_contract.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_e_scale.txt
python tools/longread_insert_ab.py 1500000 RB_GROUP_IDX=0 > $O/r04_e_longread_ab.txt 2>&1
for i in 1 2; do
RB_GROUP_IDX=0 python bench.py --no-cpu-baseline > $O/r04_e_bench_idx0_$i.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/r04_e_bench_idx1_$i.json 2>/dev/null
done
