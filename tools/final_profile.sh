#!/bin/bash
# Collects everything profiles/ holds for a round (run through gpurun from the repo root):  tools/final_profile.sh r02
#   GPU test suite, default bench line, rocprofv3 kernel stats, PMC FETCH_SIZE / WRITE_SIZE passes (each in its own run),
#   serial stage times, FETCH_SIZE calibration on k_part_count (known byte count).
#   second argument: the commit the tree was at (the GPU box has no .git):  tools/final_profile.sh r05 $(git rev-parse HEAD)
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
HEAD_SHA=${2:-unrecorded}
OUT=$R/gpurun_out/final_$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-leg > $OUT/bench_pmc_$c.json 2> $OUT/rocprof_$c.err
  python $R/profiles/summarize.py pmc $(find /tmp/prof_$c -name '*counter_collection.csv' | head -1) $c > $OUT/pmc_$c.csv
done
# which code the counter passes ran on: the id compiled into the library (= tools/csrc_id.py of the tree) and the commit; bench.py quotes
# the counters only while its library carries the same id
python - <<PY > $OUT/pmc_meta.json
import json, sys
sys.path[:0] = ["$R", "$R/rna-bloom_amd", "$R/tools"]
from rnabloom import _native as N
import csrc_id
lib, src = N.lib.rb_build_id().decode(), csrc_id.csrc_id("$R")
assert lib == src, "librb_hip.so was not built from this tree: %s vs %s" % (lib, src)
print(json.dumps({"csrc_id": lib, "git_head": "$HEAD_SHA", "run_steps": 2, "command": "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-leg under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"}))
PY
# the bench lines come AFTER the counter passes: bench.py quotes counter bytes only from summaries whose csrc id is its library's, so this round's summaries
# go where it looks (the box's copy of profiles/) first
cp $OUT/pmc_FETCH_SIZE.csv $R/profiles/${TAG}_pmc_fetch_size.csv; cp $OUT/pmc_WRITE_SIZE.csv $R/profiles/${TAG}_pmc_write_size.csv; cp $OUT/pmc_meta.json $R/profiles/${TAG}_pmc_meta.json
(cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err; RB_SERIAL=1 python bench.py --no-cpu-baseline > $OUT/bench_serial_stages.json 2> $OUT/bench_serial.err)
python - <<PY > $OUT/pmc_calibration.txt
import json
b = json.load(open("$OUT/bench_pmc_FETCH_SIZE.json"))
steps = b["steps"] + b["warmup"]
sorted_total = b["config"]["sorted_kmers_per_step"] * steps
rows = [l.rsplit(",", 3) for l in open("$OUT/pmc_FETCH_SIZE.csv").read().splitlines()[1:]]
hit = [r for r in rows if r[0].startswith("rb::k_part_count")]
fetch = sum(float(r[2]) for r in hit) * 1024
known = 8.0 * 2 * sorted_total       # the histogram kernel reads the 8-byte key of every record, once per partition pass (2 passes)
print("FETCH_SIZE calibration on rb::k_part_count (8-byte-per-lane coalesced streaming reads, nothing else read but %d B of tile descriptors):" % 16)
print("  records per step %d x %d steps x 2 passes x 8 B = %.3f GB read by construction" % (b["config"]["sorted_kmers_per_step"], steps, known / 1e9))
print("  FETCH_SIZE total over its %d dispatches        = %.3f GB" % (sum(int(r[1]) for r in hit), fetch / 1e9))
print("  known / counter = %.3f   (the guide: 128-byte requests of coalesced streaming reads are tallied as 64 bytes -> x 2; since round 4 the kernel name also covers the LSD passes of the conflict path's small sorts, whose reads are in the counter and not in the known bytes: 2.00 in rounds 2-3, a few percent lower now)" % (known / fetch))
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_cal -o c -- $R/tools/microbench/gather_bench calib > $OUT/calib.log 2>&1
python - <<PY >> $OUT/pmc_calibration.txt
import subprocess, sys
rows = subprocess.run([sys.executable, "$R/profiles/summarize.py", "pmc", subprocess.run("find /tmp/prof_cal -name '*counter_collection.csv' | head -1", shell=True, capture_output=True, text=True).stdout.strip(), "FETCH_SIZE"], capture_output=True, text=True).stdout.splitlines()[1:]
print("FETCH_SIZE per random request (tools/microbench/gather_bench calib: 2^28 requests per kernel into a 4 GB table, one lane each):")
for r in rows:
    k, n, tot, _ = r.rsplit(",", 3)
    if k.startswith("k_calib"):
        print("  %-14s reads %3s bytes per request: FETCH_SIZE %.3f GB = %.1f bytes per request" % (k, k[k.index("<") + 1:-1], float(tot) * 1024 / 1e9, float(tot) * 1024 / 2 ** 28))
print("  -> a 128-byte bucket read is tallied as 64 bytes like a 128-byte streaming request: x 2 for k_filter_reads' bucket fetches")
PY
cat $OUT/pytest_gpu.txt; tail -c 1500 $OUT/bench.json; echo; head -14 $OUT/kernel_stats.csv; cat $OUT/pmc_calibration.txt; head -8 $OUT/pmc_FETCH_SIZE.csv
