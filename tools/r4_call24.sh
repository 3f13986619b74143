#!/bin/bash
# round 4, call 24: BiLSTM kernels with their per-frame loads a frame ahead / stores a frame late: tests, then the probe, then the Force lines
set -e
mkdir -p gpurun_out/r4N
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_force.py tests/test_gpu_force_base.py tests/test_gpu_determinism.py -x -q -m gpu > gpurun_out/r4N/pytest.log 2>&1 || { tail -30 gpurun_out/r4N/pytest.log; exit 1; }
tail -2 gpurun_out/r4N/pytest.log
timeout -k 10 120 python tools/lstm_probe.py > gpurun_out/r4N/lstm.txt 2> gpurun_out/r4N/err.txt
cat gpurun_out/r4N/lstm.txt
for r in 1 2; do
timeout -k 10 300 python bench.py --workload force --steps 30 --warmup 10 --no-exact-line 2>> gpurun_out/r4N/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('force bf16', d['ms_per_step'], d['value'])"
done
