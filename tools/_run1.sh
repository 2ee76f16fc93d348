RB_DEBUG=1 timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/bench_dbg.log 2>&1
grep "\[rb\]" gpurun_out/bench_dbg.log | grep -v "N=.*D=" | tail -64
