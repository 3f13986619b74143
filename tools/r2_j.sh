#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2j
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
P=${1:-0.0}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$O/p1" -- python3 "$R/tools/attn_probe.py" $P > "$O/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/p2" -- python3 "$R/tools/attn_probe.py" $P > "$O/p2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/st" -- python3 "$R/tools/attn_probe.py" $P > "$O/st.log" 2>&1
python3 "$R/tools/pmc_kernels.py" "$O/p1" "$O/p2" > "$O/pmc.json"
cat "$O/pmc.json"
grep attn "$O"/st/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
