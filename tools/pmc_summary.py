#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, as MI355X_MICROARCH.md prescribes) into
profiles/<name>.json: average per-launch HBM-side traffic per kernel.  gfx950 correction: FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read -> reads = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Units of both counters: KiB."""
import collections, csv, glob, json, sys

def load(pattern):
    by = collections.defaultdict(list)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return by

fetch_glob, write_glob, out = sys.argv[1], sys.argv[2], sys.argv[3]
f, w = load(fetch_glob), load(write_glob)
res = {}
for k in sorted(f):
    if not any(t in k for t in ("anonymous namespace", "gemm", "attn")):
        continue
    n = len(f[k]); fe = sum(f[k]) / n
    ww = w.get(k, [])
    wr = sum(ww) / len(ww) if ww else 0.0
    res[k] = {"launches_in_trace": n, "FETCH_SIZE_KiB_avg": round(fe, 1), "WRITE_SIZE_KiB_avg": round(wr, 1),
              "hbm_bytes_per_launch_corrected": int((2 * fe + wr) * 1024)}
json.dump({"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over "
                     "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager`; reads = 2 x FETCH_SIZE (gfx950), "
                     "writes = WRITE_SIZE; Infinity-Cache hits are counted too, so re-reads beyond L2 show up here",
           "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
