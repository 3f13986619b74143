#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4l; mkdir -p "$O"; cd "$R"
timeout -k 10 300 python tools/gemm_vs_lib.py > "$O/gemm_vs_library.txt" 2> "$O/err.txt" || { tail -20 "$O/err.txt"; exit 1; }
cat "$O/gemm_vs_library.txt"
timeout -k 10 300 python bench.py --model large --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_large.json" 2>> "$O/err.txt" || { tail -20 "$O/err.txt"; exit 1; }
cut -c1-200 "$O/bench_large.json"
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py -m gpu -q -x > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -2 "$O/pytest.log"
timeout -k 10 600 python bench.py --workload force --steps 10 --warmup 3 --no-cpu-baseline > "$O/force.json" 2>> "$O/err.txt" || { tail -20 "$O/err.txt"; exit 1; }
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4l/force.json").read().strip().splitlines()[-1])
print("force", d["value"], d["ms_per_step"], {k:v for k,v in d.get("index_exact",{}).items() if k!="note"})
PY
