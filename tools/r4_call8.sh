#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4h; mkdir -p "$O"; cd "$R"
timeout -k 10 300 python tools/wgrad_probe.py > "$O/wgrad_probe.txt" 2> "$O/err.txt" || { tail -20 "$O/err.txt"; exit 1; }
cat "$O/wgrad_probe.txt"
