#!/bin/bash
# round 4, call 1: diagnosis of the red round-3 test + the tests that cover the order-fixed sums and the conv0 bounds
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4a; mkdir -p "$O"; cd "$R"
timeout -k 10 400 python tools/pr_graph_vs_eager.py > "$O/diag.log" 2>&1 || { tail -40 "$O/diag.log"; exit 1; }
tail -5 "$O/diag.log"
timeout -k 10 600 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_conv0_bwd.py tests/test_gpu_ctc_pr.py tests/test_gpu_force.py tests/test_gpu_graphed.py tests/test_gpu_train_loops2.py -m gpu -q > "$O/pytest.log" 2>&1 || { tail -60 "$O/pytest.log"; exit 1; }
tail -3 "$O/pytest.log"
