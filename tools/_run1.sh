RB_SHARD_DRIVER=native timeout 900 python bench.py --force-sharded --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-600
timeout 900 python bench.py --force-sharded --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-400
