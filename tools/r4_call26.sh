#!/bin/bash
# round 4, call 26: BiLSTM kernels holding their compute units alone (unused LDS request) beside the pipelined encoder pass
set -e
mkdir -p gpurun_out/r4E
cd /root/repo
for r in 1 2; do
 for kb in 0 120; do
  APTAI_LSTM_LDS_KB=$kb timeout -k 10 300 python bench.py --workload force --steps 30 --warmup 10 --no-exact-line 2>> gpurun_out/r4E/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lds_kb=$kb force bf16', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4E/ab.txt
 done
done
for kb in 0 120; do
  APTAI_LSTM_LDS_KB=$kb timeout -k 10 300 python bench.py --workload force --encoder-precision f32x3 --steps 12 --warmup 4 --no-exact-line 2>> gpurun_out/r4E/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lds_kb=$kb force f32x3', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4E/ab.txt
done
