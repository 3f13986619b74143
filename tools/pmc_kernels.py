"""Fold rocprofv3 --pmc counter_collection CSVs (one directory per pass) into a per-kernel table of mean counter values."""
import csv
import glob
import os
import json
import re
import sys
from collections import defaultdict


def main():
    out = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        for f in files[-1:]:                                   # the newest run only (the directories accumulate runs)
            for r in csv.DictReader(open(f)):
                m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
                name = m.group(1) if m else r["Kernel_Name"][:60]
                out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    table = {}
    for k, cs in out.items():
        if "attn" not in k and len(sys.argv) > 0 and "--all" not in sys.argv:
            continue
        table[k] = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}     # first launch dropped (cold)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
