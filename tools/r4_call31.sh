#!/bin/bash
# round 4, call 31: encoder tile for the large / 30 s Force step (BASELINE configs[4]) and the exact mode, BiLSTM isolated
set -e
mkdir -p gpurun_out/r4J
cd /root/repo
for tile in 0 128; do
 for prec in bf16 mxfp8; do
  APTAI_FORCE_ENC_TILE=$tile timeout -k 10 400 python bench.py --workload force --model large --seconds 30 --encoder-precision $prec --steps 10 --warmup 3 --no-exact-line --no-cpu-baseline 2>> gpurun_out/r4J/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('enc_tile=$tile large 30 s $prec', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4J/ab.txt
 done
 APTAI_FORCE_ENC_TILE=$tile timeout -k 10 300 python bench.py --workload force --encoder-precision f32x3 --steps 12 --warmup 4 --no-exact-line --no-cpu-baseline 2>> gpurun_out/r4J/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('enc_tile=$tile base f32x3', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4J/ab.txt
done
