#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -5 > $O/r04_ab_tests.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_ab_bench_rocprof.json 2>/dev/null
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $O/r04_ab_kernel_stats.csv
cd $R
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_ab_bench$i.json; done
RB_PART_PERSIST=0 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_ab_bench_nopersist.json
