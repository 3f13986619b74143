#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4j; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_determinism.py tests/test_gpu_force.py -m gpu -q -x -s > "$O/pytest.log" 2>&1 || { tail -60 "$O/pytest.log"; exit 1; }
grep -E "bands|passed|failed" "$O/pytest.log" | tail -12
for prec in f32x3 f32x6; do
timeout -k 10 400 python bench.py --workload force --encoder-precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-exact-line > "$O/force_$prec.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
cut -c1-170 "$O/force_$prec.json"
done
