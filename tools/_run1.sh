set -x
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_cal -o c -- $R/tools/microbench/gather_bench calib > $R/gpurun_out/calib.log 2>&1
python $R/profiles/summarize.py pmc $(find /tmp/prof_cal -name '*counter_collection.csv' | head -1) FETCH_SIZE > $R/gpurun_out/calib_fetch.csv
cat $R/gpurun_out/calib_fetch.csv
