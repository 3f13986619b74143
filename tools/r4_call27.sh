#!/bin/bash
set -e
mkdir -p gpurun_out/r4F
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_env_knobs.py tests/test_gpu_force.py tests/test_gpu_force_base.py -x -q -m gpu > gpurun_out/r4F/pytest.log 2>&1 || { tail -30 gpurun_out/r4F/pytest.log; exit 1; }
tail -2 gpurun_out/r4F/pytest.log
for kb in 136 120 0; do
  APTAI_LSTM_LDS_KB=$kb timeout -k 10 300 python bench.py --workload force --steps 30 --warmup 10 --no-exact-line 2>> gpurun_out/r4F/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lds_kb=$kb force bf16', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4F/ab.txt
done
