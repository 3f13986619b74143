#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2m
mkdir -p "$O"
cd "$R"
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for pr in 0 1; do
  APTAI_FORCE_HEADS_PRIORITY=$pr timeout -k 10 300 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force_$pr.json" 2> "$O/force.err" || { tail -30 "$O/force.err"; exit 1; }
  echo "heads priority $pr $(cut -c1-200 "$O/force_$pr.json")"
done
timeout -k 10 600 python -m pytest tests/test_gpu_force.py -m gpu -q -x -k "prefetched or golden or config3" 2>&1 | tail -3
