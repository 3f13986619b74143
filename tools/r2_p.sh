#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2p
mkdir -p "$O"
cd "$R"
for m in ffffffff fefefefe eeeeeeee; do
  APTAI_FORCE_ENC_CUMASK=$m timeout -k 10 120 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force_$m.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
  echo "mask $m $(cut -c1-200 "$O/force_$m.json")"
done
APTAI_FORCE_ENC_CUMASK=fefefefe PIPE=1 timeout -k 10 120 python tools/force_timeline.py 2>&1 | grep -v amdgpu.ids | tail -12
