#!/bin/bash
set -e
mkdir -p gpurun_out/r4p
cd /root/repo
for v in 0 1; do
  APTAI_ADAM_OVERLAP=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-exact-line 2> gpurun_out/r4p/err_$v.txt > gpurun_out/r4p/out_$v.json
  python -c "
import json
d = json.loads([l for l in open('gpurun_out/r4p/out_$v.json') if l.startswith('{')][-1]); r = d['roofline']
print('overlap=$v', d['ms_per_step'], r['frac'], r['launches'], r['avg_launch_us'], r['measured'])"
done
