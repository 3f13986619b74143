#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2q
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_force.py -m gpu -q -x > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -2 "$O/pytest.log"
for sp in 0 1; do
  APTAI_FORCE_ENC_GRAPH=$sp timeout -k 10 120 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force_$sp.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
  echo "encoder graph $sp $(cut -c100-200 "$O/force_$sp.json")"
done
