set -x
export TMPDIR=/tmp
timeout 1200 python tools/measure_host_path.py 10000000 > gpurun_out/host_path.log 2>&1
R=$PWD
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loop8 -o loop8 --output-format csv -- python $R/tools/loopback_bench.py --ranks 8 --pairs 50000000 --warmup 0 --steps 1 > $R/gpurun_out/loop8_prof.log 2>&1
cd $R
find gpurun_out/prof_loop8 -name "*kernel_stats*" | head
find gpurun_out/prof_loop8 -name "*kernel_trace*" -delete
