#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_walks.py tests/test_gpu_scale.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_o_tests.txt
for i in 1; do
RB_SERIAL=1 RB_GROUP_CLASSES=0 python bench.py --no-cpu-baseline > $O/r04_o_serial_cls0_$i.json 2>/dev/null
RB_SERIAL=1 python bench.py --no-cpu-baseline > $O/r04_o_serial_cls1_$i.json 2>/dev/null
RB_GROUP_CLASSES=0 python bench.py --no-cpu-baseline > $O/r04_o_bench_cls0_$i.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/r04_o_bench_cls1_$i.json 2>/dev/null
done
