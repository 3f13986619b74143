#!/usr/bin/env python3
"""Summarise one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass into profiles/<name>.json: per kernel, the
fraction of SIMD cycles in which the matrix pipe was busy.  SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over the 1024 SIMDs
(MI355X_MICROARCH.md: = 32 x N for v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE counts GPU-active cycles of the dispatch (summed over
the 8 XCDs by rocprofv3, hence / 8).  mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import collections, csv, json, sys
src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(src)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy, act = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"])
    if busy <= 0 or act <= 0:
        continue
    res[k] = {"launches_in_trace": len(c["GRBM_GUI_ACTIVE"]), "mfma_busy_cycles_avg": round(busy / len(c["GRBM_GUI_ACTIVE"])),
              "gui_active_cycles_avg": round(act / len(c["GRBM_GUI_ACTIVE"])), "mfma_util": round(busy / (act / 8 * 1024), 4)}
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --eager`; mfma_util = busy cycles / (GPU-active cycles x 1024 SIMDs), GUI_ACTIVE summed over 8 XCDs",
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_avg"] * kv[1]["launches_in_trace"]))}, open(out, "w"), indent=1)
for k, v in list(json.load(open(out))["kernels"].items())[:14]:
    print(f"{k[:70]:70s} util {v['mfma_util']:.3f}  active {v['gui_active_cycles_avg']}")
