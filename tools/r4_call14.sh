#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4n; mkdir -p "$O"; cd "$R"
APTAI_HIP_LIB=$R/tools/ab/stag/lib_stagger.so timeout -k 10 300 python tools/stagger_probe.py > "$O/stagger.txt" 2> "$O/err.txt" || { tail -20 "$O/err.txt"; exit 1; }
cat "$O/stagger.txt"
