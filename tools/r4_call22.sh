#!/bin/bash
# round 4, call 22: timing experiment - the four dgrads of a layer as NT launches on (stale) transposed weight copies
set -e
mkdir -p gpurun_out/r4v
cd /root/repo
for r in 1 2 3; do
  for v in 0 1; do
    APTAI_EXP_DGRAD_NT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-exact-line 2> gpurun_out/r4v/err_$v.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('dgrad_nt=$v', 'ms_per_step', d['ms_per_step'], 'loss', d['loss'])
" | tee -a gpurun_out/r4v/ab.txt
  done
done
