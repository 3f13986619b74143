#!/bin/bash
# Regenerates the artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/refresh_profiles.sh r01'
# 1. kernel-trace statistics of the default bench run, 2.-4. PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) over a short
# eager run of the same step (one counter set per pass, --kernel-trace only, as the pool requires), 5. the bench line.
# Outputs land in gpurun_out/profiles_<tag>/ and are folded into profiles/<tag>_* by the tools/pmc_*.py scripts afterwards.
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_$TAG
mkdir -p "$O"
# (each pass writes into a fresh directory: rocprofv3 adds files next to older ones)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/stats.log" 2>&1
echo "[refresh] stats done"
for pass in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$O/$pass" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --eager > "$O/$pass.log" 2>&1
    echo "[refresh] $pass done"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/mfma" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --eager > "$O/mfma.log" 2>&1
echo "[refresh] mfma done"
cd "$R" && python3 bench.py > "$O/bench.json" 2> "$O/bench.log"
tail -1 "$O/bench.json" | cut -c1-300
