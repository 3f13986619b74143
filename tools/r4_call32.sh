#!/bin/bash
# round 4, call 32: BiLSTM isolation on / off at the large / 30 s Force step (encoder-bound)
set -e
mkdir -p gpurun_out/r4K
cd /root/repo
for r in 1 2; do
for kb in 136 0; do
 for prec in bf16 mxfp8; do
  APTAI_LSTM_LDS_KB=$kb timeout -k 10 400 python bench.py --workload force --model large --seconds 30 --encoder-precision $prec --steps 10 --warmup 3 --no-exact-line --no-cpu-baseline 2>> gpurun_out/r4K/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lds_kb=$kb large 30 s $prec', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4K/ab.txt
 done
done
done
