#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2v
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for m in 8 0 4; do
  APTAI_GEMM_RASTER=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/r$m" -- python3 "$R/bench.py" ${BENCH_ARGS} --steps 8 --warmup 3 --no-cpu-baseline > "$O/r$m.log" 2>&1
  echo "raster $m done: $(grep -o '"ms_per_step": [0-9.]*' "$O/r$m.log")"
done
