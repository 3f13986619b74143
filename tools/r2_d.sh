#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2d
mkdir -p "$O"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1 || { grep -E "FAILED|Error|assert" "$O/pytest.log" | head -60; exit 1; }
grep -E "passed|failed" "$O/pytest.log" | tail -3
timeout -k 10 300 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
cut -c1-260 "$O/force.json"
timeout -k 10 300 python bench.py --workload pr --steps 10 --warmup 3 --no-cpu-baseline > "$O/pr.json" 2> "$O/pr.err" || { tail -30 "$O/pr.err"; exit 1; }
cut -c1-260 "$O/pr.json"
for r in 0 8 0 8; do
  APTAI_GEMM_RASTER=$r timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$O/aptai_raster$r.json" 2> "$O/aptai.err" || { tail -30 "$O/aptai.err"; exit 1; }
  echo "raster=$r $(cut -c1-200 "$O/aptai_raster$r.json")"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_stats.log" 2>&1
echo "[r2d] force stats done"
