#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4b; mkdir -p "$O"; cd "$R"
: > "$O/cause.txt"
for lib in "" tools/ab/r4cause/lib_r3.so tools/ab/r4cause/lib_newctc_oldconv.so tools/ab/r4cause/lib_oldctc_newconv.so tools/ab/r4cause/lib_r3.so; do
  if [ -n "$lib" ]; then export APTAI_HIP_LIB=$R/$lib; else unset APTAI_HIP_LIB; fi
  timeout -k 10 200 python tools/pr_loop_ab.py 2> "$O/err.log" | grep -v amdgpu.ids >> "$O/cause.txt" || { tail -20 "$O/err.log"; exit 1; }
done
cat "$O/cause.txt"
