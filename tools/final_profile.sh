# Collects everything profiles/ holds for a round (run through gpurun from the repo root):
#   GPU test suite, default bench line, rocprofv3 kernel stats, PMC FETCH_SIZE / WRITE_SIZE passes, serial stage times.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
RB_SERIAL=1 python bench.py --no-cpu-baseline > $OUT/bench_serial_stages.json 2> $OUT/bench_serial.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_$c.json 2> $OUT/rocprof_$c.err
  python $R/profiles/summarize.py pmc $(find /tmp/prof_$c -name '*counter_collection.csv' | head -1) $c > $OUT/pmc_$c.csv
done
cat $OUT/pytest_gpu.txt; tail -c 600 $OUT/bench.json; head -12 $OUT/kernel_stats.csv; head -6 $OUT/pmc_FETCH_SIZE.csv
