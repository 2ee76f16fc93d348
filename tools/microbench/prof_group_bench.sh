#!/bin/bash
# per-kernel times of the grouping microbench: prof_group_bench.sh <n> [env assignments...]
cd /tmp && export TMPDIR=/tmp
N=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/gb_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gb_prof -o gb -- /root/repo/tools/microbench/group_bench $N 5.4 0 2>&1 | grep "^n="
find /tmp/gb_prof -name "*kernel_stats*" | head -1 | xargs cat | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'rocclr' in r[0] or 'k_fill' in r[0]: continue
    print(r[0][:70].ljust(70), r[1], r[3][:10])
"
