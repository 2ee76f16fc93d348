#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_primitives.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5 > $O/r04_p_prims.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_p_bench_rocprof.json 2>/dev/null
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $O/r04_p_kernel_stats.csv
cd $R; python bench.py --no-cpu-baseline > $O/r04_p_bench.json 2>/dev/null
