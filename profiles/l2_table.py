#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (counter_collection CSVs): one row per kernel, one column per counter, per zone.
usage: l2_table.py <dir> [<dir> ...] <zones>"""
import csv, glob, sys
from collections import defaultdict
zones = float(sys.argv[-1])
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:-1]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("| kernel | launches | " + " | ".join(n + " /zone" for n in names) + " |")
print("|---|---|" + "---|" * len(names))
rows = []
for k, v in acc.items():
    tot = sum(sum(x) / len(x) for x in v.values())
    rows.append((tot, k, v))
for tot, k, v in sorted(rows, reverse=True)[:14]:
    n = max(len(x) for x in v.values())
    print(f"| `{k[:60]}` | {n} | " + " | ".join(f"{sum(v[c]) / len(v[c]) / zones:.3f}" if c in v else "-" for c in names) + " |")
