#!/bin/bash
# round 4, call 30: with the BiLSTM alone on its compute units, which tile should the side-stream encoder be captured with?
set -e
mkdir -p gpurun_out/r4I
cd /root/repo
for r in 1 2; do
 for tile in 128 0 192; do
  APTAI_FORCE_ENC_TILE=$tile timeout -k 10 300 python bench.py --workload force --steps 30 --warmup 10 --no-exact-line 2>> gpurun_out/r4I/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('enc_tile=$tile force bf16', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4I/ab.txt
 done
done
