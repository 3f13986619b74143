#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2t
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_ctc_pr.py -m gpu -q -x > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -2 "$O/pytest.log"
for flag in "--eager" "" ""; do
  timeout -k 10 300 python bench.py --workload pr $flag --steps 20 --warmup 5 --no-cpu-baseline > "$O/pr$flag.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
  echo "pr $flag $(cut -c100-200 "$O/pr$flag.json")"
done
