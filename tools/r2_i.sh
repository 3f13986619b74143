#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2i
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_norm_attn.py tests/test_gpu_aptai.py -m gpu -q -x > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -2 "$O/pytest.log"
timeout -k 10 300 python tools/attn_sweep.py 2>&1 | tee "$O/sweep.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || { tail -20 "$O/bench.err"; exit 1; }
cut -c1-260 "$O/bench.json"
