#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4f; mkdir -p "$O"; cd "$R"
APTAI_HIP_LIB=$R/tools/ab/t4/lib_t4_stamps.so timeout -k 10 300 python tools/gemm_t4_stamps.py > "$O/t4_stamps.txt" 2> "$O/t4.err" || { tail -20 "$O/t4.err"; exit 1; }
cat "$O/t4_stamps.txt"
