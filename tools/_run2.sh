set -x
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loop2 -o loop2 --output-format csv -- python $R/tools/loopback_bench.py --ranks 2 --pairs 50000000 --warmup 0 --steps 1 > $R/gpurun_out/loop2_prof.log 2>&1
cd $R
find gpurun_out/prof_loop2 -name "*kernel_trace*" -delete
