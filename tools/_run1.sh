set -x
timeout 900 python -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/pytest_fq.log 2>&1
tail -15 gpurun_out/pytest_fq.log
export TMPDIR=/tmp
RB_HOST_TIMING=1 timeout 1200 python tools/measure_host_path.py 10000000 > gpurun_out/host_path.log 2>&1
grep -v "^+" gpurun_out/host_path.log | tail -8
