#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_config3.py tests/test_gpu_sharded.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_h_sharded.txt
for v in 96 48 24 12; do
RB_SERIAL=1 RB_LIGHT_OPS=$v python bench.py --no-cpu-baseline > $O/r04_h_bench_light$v.json 2>/dev/null
done
