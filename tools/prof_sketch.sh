# kernel-level rates of the sketch calls: one piece of the long-read tool under rocprofv3 --kernel-trace --stats
# usage (through gpurun): bash tools/prof_sketch.sh [reads=250000]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-250000}
mkdir -p $R/gpurun_out
RB_LR_SKIP_INSERT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sketch -o p -- python $R/tools/longread_full.py $N $N > $R/gpurun_out/prof_sketch.log 2>&1
f=$(find /tmp/prof_sketch -name '*kernel_stats.csv' | head -1)
python - "$f" > $R/gpurun_out/prof_sketch_kernels.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("kernel,calls,total_ms,avg_ms")
for r in rows[:25]:
    print('"%s",%s,%.3f,%.3f' % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
tail -6 $R/gpurun_out/prof_sketch.log; head -14 $R/gpurun_out/prof_sketch_kernels.csv
