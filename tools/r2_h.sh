#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2h
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_ctc_pr.py tests/test_gpu_force.py -m gpu -q -s -k "golden" > "$O/bands.log" 2>&1 || true
grep -E "bands\]|passed|failed" "$O/bands.log" | tail -60
for prec in bf16 mxfp8; do
  timeout -k 10 400 python bench.py --workload force --model large --seconds 30 --batch 4 --encoder-precision $prec --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_large30_$prec.json" 2> "$O/force_l.err" || { tail -30 "$O/force_l.err"; exit 1; }
  echo "large-30s $prec $(cut -c1-200 "$O/force_large30_$prec.json")"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 2 --no-cpu-baseline > "$O/force_stats.log" 2>&1
echo "[r2h] force stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pr_stats" -- python3 "$R/bench.py" --workload pr --steps 5 --warmup 2 --no-cpu-baseline > "$O/pr_stats.log" 2>&1
echo "[r2h] pr stats done"
