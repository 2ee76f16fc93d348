timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -k "get_kmers or walks" 2>&1 | tail -15
