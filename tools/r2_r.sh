#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2r
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_norm_attn.py tests/test_gpu_gemm.py tests/test_gpu_aptai.py tests/test_gpu_ctc_pr.py tests/test_gpu_graphed.py -m gpu -q -x -s > "$O/pytest.log" 2>&1 || { grep -E "FAILED|Error|assert|error" "$O/pytest.log" | head -40; exit 1; }
grep -E "passed|failed|bands\]" "$O/pytest.log" | tail -8
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/aptai.json" 2> "$O/aptai.err" || { tail -30 "$O/aptai.err"; exit 1; }
cut -c1-230 "$O/aptai.json"
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-230 "$O/pr.json"
