#!/bin/bash
# round 4, call 17: optimiser-overlap equality test, the optimiser tests, and the bench line with the plugged probe (piped, as call 15 ran it)
set -e
mkdir -p gpurun_out/r4q
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_optim.py -x -q -m gpu > gpurun_out/r4q/pytest.log 2>&1 || { tail -30 gpurun_out/r4q/pytest.log; exit 1; }
tail -3 gpurun_out/r4q/pytest.log
for r in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-exact-line 2> gpurun_out/r4q/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('ms_per_step', d['ms_per_step'], 'frac', r['frac'], r['launches'], r['avg_launch_us'])
" | tee -a gpurun_out/r4q/ab.txt
done
