#!/bin/bash
# round 4, call 20: kernel statistics of the exact-index Force_APTAI step as it stands (f32x3)
set -e
mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4t/stats -o runc -- python3 /root/repo/bench.py --workload force --encoder-precision f32x3 --steps 12 --warmup 4 --no-exact-line > /root/repo/gpurun_out/r4t/bench.json 2> /root/repo/gpurun_out/r4t/err.txt
tail -2 /root/repo/gpurun_out/r4t/bench.json | cut -c1-300
