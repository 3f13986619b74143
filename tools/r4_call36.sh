#!/bin/bash
# round 4, call 36: Force_APTAI bench records on the final library (base 10 s: bf16 + its f32x3 line, f32x6; large 30 s: bf16, mxfp8)
set -e
O=gpurun_out/r4R
mkdir -p $O
cd /root/repo
timeout -k 10 400 python bench.py --workload force --steps 30 --warmup 10 > $O/bench_force.json 2> $O/err1.txt
timeout -k 10 400 python bench.py --workload force --encoder-precision f32x6 --steps 12 --warmup 4 --no-exact-line > $O/bench_force_f32x6.json 2> $O/err2.txt
for prec in bf16 mxfp8; do
  timeout -k 10 400 python bench.py --workload force --model large --seconds 30 --encoder-precision $prec --steps 10 --warmup 3 --no-exact-line > $O/bench_force_large30_$prec.json 2> $O/err3.txt
done
timeout -k 10 400 python bench.py --workload pr --steps 20 --warmup 5 > $O/bench_pr.json 2> $O/err4.txt
for f in $O/*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['value'], (d.get('index_exact') or {}).get('ms_per_step'))"; done
