#!/bin/bash
# round 4, call 28: kernel trace of the Force_APTAI bf16 step as it stands
set -e
mkdir -p gpurun_out/r4M
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4M/stats -o runc -- python3 /root/repo/bench.py --workload force --steps 12 --warmup 4 --no-exact-line > /root/repo/gpurun_out/r4M/bench.json 2> /root/repo/gpurun_out/r4M/err.txt
tail -1 /root/repo/gpurun_out/r4M/bench.json | cut -c1-200
