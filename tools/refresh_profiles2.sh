#!/bin/bash
# Round-2 additions to tools/refresh_profiles.sh: kernel statistics of the force / pr workloads and the attention PMC passes.
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02 && bash tools/refresh_profiles2.sh r02'
set -eo pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_$TAG
mkdir -p "$O"
# (each pass writes into a fresh directory: rocprofv3 adds files next to older ones)
cd /tmp && export TMPDIR=/tmp
for wl in force pr; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${wl}_stats" -- python3 "$R/bench.py" --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > "$O/${wl}_stats.log" 2>&1
    echo "[refresh2] $wl stats done"
done
for p in 0.0 0.1; do
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$O/attn_p${p}_a" -- python3 "$R/tools/attn_probe.py" $p > "$O/attn_a.log" 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/attn_p${p}_b" -- python3 "$R/tools/attn_probe.py" $p > "$O/attn_b.log" 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/attn_p${p}_st" -- python3 "$R/tools/attn_probe.py" $p > "$O/attn_st.log" 2>&1
    echo "[refresh2] attention p=$p done"
done
cd "$R"
for wl in force pr; do
    python3 bench.py --workload $wl > "$O/bench_$wl.json" 2> "$O/bench_$wl.log"
    tail -1 "$O/bench_$wl.json" | cut -c1-200
done
