#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4e; mkdir -p "$O"; cd "$R"
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k "256x192" > "$O/pytest.log" 2>&1 || { tail -40 "$O/pytest.log"; exit 1; }
tail -2 "$O/pytest.log"
timeout -k 10 500 python tools/gemm_t4_bench.py --rounds 3 --tiles 128,448 > "$O/t4_plain.txt" 2> "$O/t4.err" || { tail -20 "$O/t4.err"; exit 1; }
cat "$O/t4_plain.txt"
APTAI_HIP_LIB=$R/tools/ab/t4/lib_t4_stamps.so timeout -k 10 300 python tools/gemm_t4_stamps.py > "$O/t4_stamps.txt" 2>> "$O/t4.err" || { tail -20 "$O/t4.err"; exit 1; }
cat "$O/t4_stamps.txt"
