#!/bin/bash
# Full GPU pass on the box: pytest -m gpu, then the three bench workloads (gpurun --timeout 1200 -- bash tools/gpu_suite.sh)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/gpu_suite
mkdir -p "$O"
cd "$R"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1 || { grep -E "FAILED|Error|assert|error" "$O/pytest.log" | head -60; exit 1; }
grep -E "passed|failed" "$O/pytest.log" | tail -3
for prec in bf16 mxfp8; do
  timeout -k 10 300 python bench.py --workload force --encoder-precision $prec --steps 10 --warmup 3 --no-cpu-baseline > "$O/force_$prec.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
  echo "$prec $(cut -c1-200 "$O/force_$prec.json")"
done
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-260 "$O/pr.json"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/aptai.json" 2> "$O/aptai.err" || { tail -30 "$O/aptai.err"; exit 1; }
cut -c1-260 "$O/aptai.json"
