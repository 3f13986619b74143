#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2e
mkdir -p "$O"
cd "$R"
timeout -k 10 1100 python -m pytest tests/test_gpu_train_loops2.py tests/test_gpu_config5.py tests/test_gpu_ctc_pr.py tests/test_gpu_aptai.py -m gpu -q -x -s > "$O/pytest.log" 2>&1 || { grep -E "margin\]|FAILED|Error|assert" "$O/pytest.log" | head -60; exit 1; }
grep -E "margin\]|passed|failed" "$O/pytest.log" | tail -8
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-260 "$O/pr.json"
