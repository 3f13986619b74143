#!/bin/bash
# round 4, call 15: optimiser under the backward pass (APTAI_ADAM_OVERLAP), A/B on one box + the tests that pin the step
set -e
mkdir -p gpurun_out/r4o
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_optim.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4o/pytest.log 2>&1 || { tail -30 gpurun_out/r4o/pytest.log; exit 1; }
tail -3 gpurun_out/r4o/pytest.log
for r in 1 2 3; do
  for v in 0 1; do
    APTAI_ADAM_OVERLAP=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-exact-line 2> gpurun_out/r4o/err_$v.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('overlap=$v', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'])
" | tee -a gpurun_out/r4o/ab.txt
  done
done
