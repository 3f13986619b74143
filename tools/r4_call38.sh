#!/bin/bash
# round 4, call 38: PMC evidence for "the attention kernels are bound by their vector issue": VALU-active and MFMA-busy cycles per kernel
set -e
R=/root/repo
O=$R/gpurun_out/r4T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/valu -- python3 $R/tools/attn_bench.py > $O/valu.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/tools/attn_bench.py > $O/mfma.log 2>&1
ls $O/valu/*/ $O/mfma/*/ | head
