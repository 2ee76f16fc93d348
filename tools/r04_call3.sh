#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_primitives.py -x -q 2>&1 | tail -5 > $O/r04_c_prims.txt
timeout 900 tools/microbench/alloc_explain > $O/r04_c_alloc.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -5 > $O/r04_c_parity.txt
python bench.py --no-cpu-baseline > $O/r04_c_bench.json 2>/dev/null
