set -x
export TMPDIR=/tmp
R=$PWD
for g in 1 2 4 8; do timeout 900 python tools/loopback_bench.py --ranks $g --pairs 50000000 --trace > gpurun_out/loopf_$g.log 2>&1; done
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loop8f -o loop8 --output-format csv -- python $R/tools/loopback_bench.py --ranks 8 --pairs 50000000 --warmup 1 --steps 1 > $R/gpurun_out/loop8f_prof.log 2>&1
cd $R
find gpurun_out/prof_loop8f -name "*kernel_trace*" -delete
