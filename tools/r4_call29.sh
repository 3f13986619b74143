#!/bin/bash
# round 4, call 29: dQ kernel (key mask peeled into the boundary tile, 2-deep K/V ring with one barrier per tile): tests, then A/B timing
set -e
mkdir -p gpurun_out/r4H
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_norm_attn.py tests/test_gpu_aptai.py -x -q -m gpu > gpurun_out/r4H/pytest.log 2>&1 || { tail -30 gpurun_out/r4H/pytest.log; exit 1; }
tail -2 gpurun_out/r4H/pytest.log
for r in 1 2; do
APTAI_HIP_LIB=$PWD/tools/ab/attn/lib_before.so timeout -k 10 200 python tools/attn_bench.py 2>>gpurun_out/r4H/err.txt | sed 's/^/before: /' | tee -a gpurun_out/r4H/attn.txt
timeout -k 10 200 python tools/attn_bench.py 2>>gpurun_out/r4H/err.txt | sed 's/^/after:  /' | tee -a gpurun_out/r4H/attn.txt
done
