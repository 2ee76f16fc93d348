set -x
RB_SHARD_MPF=1 timeout 900 python tools/loopback_bench.py --ranks 2 --pairs 50000000 --trace > gpurun_out/loop2_mpf.log 2>&1
RB_SHARD_MPF=1 timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/pytest_mpf.log 2>&1
tail -3 gpurun_out/pytest_mpf.log
