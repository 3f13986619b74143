#!/bin/bash
set -e
mkdir -p gpurun_out/r4u
cd /root/repo
timeout -k 10 300 python tools/exact_gemm_tiles.py 3 > gpurun_out/r4u/tiles3.txt 2> gpurun_out/r4u/err.txt
cat gpurun_out/r4u/tiles3.txt
