set -x
bash tools/microbench/run_group_bench.sh > gpurun_out/gb_check.log 2>&1
timeout 300 tools/microbench/group_bench 200000000 1.5 5 > gpurun_out/gb_perf.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/bench1.log 2>&1
timeout 900 python tools/loopback_bench.py --ranks 8 --pairs 50000000 --trace > gpurun_out/loop8.log 2>&1
timeout 900 python tools/loopback_bench.py --ranks 2 --pairs 50000000 --trace > gpurun_out/loop2.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest1.log 2>&1
tail -3 gpurun_out/pytest1.log
