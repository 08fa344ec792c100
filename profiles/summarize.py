#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-dispatch FETCH_SIZE / WRITE_SIZE passes)
into the per-kernel table committed under profiles/.
usage: summarize.py <trace_dir> <fetch_dir> <write_dir> <zones> > profiles/rNN_summary.md"""
import csv
import glob
import sys
from collections import defaultdict


def one(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return f[0] if f else None


def counters(d):
    acc = defaultdict(list)
    f = one(d, "*_counter_collection.csv")
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


trace, fetch, write, zones = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
fs, ws = counters(fetch), counters(write)
print("| kernel | calls | avg ms | % | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM B/zone (fetch x2 corr.) | eff. GB/s |")
print("|---|---|---|---|---|---|---|---|")
for r in csv.DictReader(open(one(trace, "*_kernel_stats.csv"))):
    n = r["Name"]
    f = sum(fs[n]) / len(fs[n]) if fs.get(n) else float("nan")
    w = sum(ws[n]) / len(ws[n]) if ws.get(n) else float("nan")
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
    # streaming reads -> doubled here; WRITE_SIZE is exact.  Units are KiB.
    byts = (2.0 * f + w) * 1024.0
    ms = float(r["AverageNs"]) / 1e6
    print(f"| `{n[:70]}` | {r['Calls']} | {ms:.3f} | {r['Percentage']} | {f:.0f} | {w:.0f} | {byts / zones:.0f} | {byts / (ms * 1e-3) / 1e9:.0f} |")

# optional 5th argument: write the per-kernel HBM bytes per launch (fetch x2 + write) keyed by the
# names bench.py's event profiler uses, for bench.py's `roofline.traffic`
if len(sys.argv) > 5:
    import json
    import re
    alias = [(r"k_ion_update", "ion_update"), (r"k_ion_rates", "ion_rates"), (r"k_ray_sweep<true>", "ray_sweep_rates"), (r"k_ray_sweep<false>", "ray_sweep"),
             (r"k_ion_begin", "ion_begin"), (r"k_cfl", "new_dt"), (r"k_update<", "update"),
             (r"k_flux2_update<", "flux2_update"), (r"k_correct_all<", "correct_all"),
             (r"k_sweep_tile<1, 1, true, 1", "correct_x2"), (r"k_sweep_tile<1, 2, true, 1", "correct_x3"),
             (r"k_sweep_x1<1, true, 3", "sweep_correct_x1"), (r"k_sweep_x1<1, true, 0", "sweep_x1"),
             (r"k_sweep_x1<1, true, 1", "correct_x1"),
             (r"k_sweep_march<1, 1, true, 0", "sweep_x2"), (r"k_sweep_march<1, 2, true, 0", "sweep_x3"),
             (r"k_slopes<", "ppm_slopes"),
             (r"k_flux2<1, 0", "flux2_x1"), (r"k_flux2<1, 1", "flux2_x2"), (r"k_flux2<1, 2", "flux2_x3")]
    out = {}
    for n in set(fs) | set(ws):
        for pat, key in alias:
            if pat in n and fs.get(n) and ws.get(n):
                out[key] = (2.0 * sum(fs[n]) / len(fs[n]) + sum(ws[n]) / len(ws[n])) * 1024.0
    json.dump({"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
                           "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section); KiB -> bytes",
               "workload": sys.argv[5], "kernels": dict(sorted(out.items()))}, open(sys.argv[6], "w"), indent=1)
