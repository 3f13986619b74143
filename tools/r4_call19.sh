#!/bin/bash
# round 4, call 19: fused exact attention - unit test, the exact-mode tests, then the Force bench lines (f32x3, f32x6)
set -e
mkdir -p gpurun_out/r4s
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_exact.py -x -q -m gpu -k "fused" -s > gpurun_out/r4s/unit.log 2>&1 || { tail -40 gpurun_out/r4s/unit.log; exit 1; }
grep -E "exact\]|passed|failed" gpurun_out/r4s/unit.log
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py -x -q -m gpu -s > gpurun_out/r4s/exact.log 2>&1 || { tail -40 gpurun_out/r4s/exact.log; exit 1; }
grep -E "exact\]|passed|failed" gpurun_out/r4s/exact.log
for prec in f32x3 f32x6; do
  timeout -k 10 300 python bench.py --workload force --encoder-precision $prec --steps 12 --warmup 4 --no-exact-line 2> gpurun_out/r4s/err_$prec.txt > gpurun_out/r4s/force_$prec.json
  python -c "
import json
d = json.loads([l for l in open('gpurun_out/r4s/force_$prec.json') if l.startswith('{')][-1]); print('$prec', d['ms_per_step'], d['value'])"
done
