#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2b
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_train_loop.py -m gpu -q -s > "$O/pytest.log" 2>&1 || { grep -E "margin\]|FAILED|Error" "$O/pytest.log" | head -40; }
grep -E "margin\]|passed|failed" "$O/pytest.log" | tail -20
timeout -k 10 300 python tools/attn_sweep.py > "$O/attn_sweep.log" 2>&1 || tail -20 "$O/attn_sweep.log"
cat "$O/attn_sweep.log" | grep "len="
timeout -k 10 300 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
cut -c1-300 "$O/force.json"
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-300 "$O/pr.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_stats.log" 2>&1
echo "[r2b] force stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pr_stats" -- python3 "$R/bench.py" --workload pr --steps 5 --warmup 2 --no-cpu-baseline > "$O/pr_stats.log" 2>&1
echo "[r2b] pr stats done"
