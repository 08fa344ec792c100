#!/usr/bin/env python3
"""Condense the rocprofv3 passes of one bench.py command into the tables committed under profiles/:
  <trace_dir>  rocprofv3 --kernel-trace --stats          (per-dispatch durations: *_kernel_trace.csv)
  <fetch_dir>  rocprofv3 --pmc FETCH_SIZE                 (separate passes: the TCC counters do not fit together)
  <write_dir>  rocprofv3 --pmc WRITE_SIZE
bench.py spins the deck up before its timed region (60-150 untimed steps in another regime), so every figure here is
taken over the LAST `steps` launches per step-kernel of the process -- the timed region -- not over the whole run.
FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE is exact; both in KiB.

usage: summarize_r02.py <trace_dir> <fetch_dir> <write_dir> <zones> <steps> <workload> <out.md> <out_traffic.json>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

trace, fetch, write, zones, steps, workload, out_md, out_json = sys.argv[1:9]


def _fp():
    """the kernel sources' fingerprint (bench.source_fingerprint): bench.py quotes these bytes only while it still matches"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_fingerprint()


zones = float(zones); steps = int(steps)

# kernel name (rocprof) -> name used by bench.py's hipEvent profiler; several kernels may share a bench name (the
# updating pass is launched in two forms of which one leaves at once)
ALIAS = [(r"k_ion_pass<true, true, false>", "ion_pass"), (r"k_ion_pass<true, false, false>", "ion_pass"),
         (r"k_ion_pass<false, true, true>", "ion_pass_begin"), (r"k_ion_pass<false, true, false>", "ion_pass_first"),
         (r"k_ion_begin16", "ion_begin"), (r"k_ion_finish", "ion_finish"), (r"k_ion_reduce", "ion_pass"), (r"k_ion_pick2", "ion_pass"),
         (r"k_ion_update", "ion_update"), (r"k_ion_rates", "ion_rates"), (r"k_ray_sweep<true>", "ray_sweep_rates"),
         (r"k_ray_sweep<false>", "ray_sweep"), (r"k_ion_begin", "ion_begin"), (r"k_cfl", "new_dt"), (r"k_update<", "update"),
         (r"k_flux2_update<", "flux2_update"), (r"k_correct_all<", "correct_all"), (r"k_eta_edges", "correct_all"), (r"k_x1_edge_flux<", "correct_all"),
         (r"k_sweep_x1<1, true, 0", "sweep_x1"), (r"k_sweep_x1_flat<1, true, 0", "sweep_x1"), (r"k_sweep_march<1, 1, true, 0", "sweep_x2"), (r"k_sweep_march<1, 2, true, 0", "sweep_x3"),
         (r"k_bc", "bvals_mhd"), (r"k_pinned", "pinned_cells")]


def alias(n):
    for pat, key in ALIAS:
        if pat in n:
            return key
    return None


def one(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return f[0] if f else None


def last_launches(rows, per_step):
    """rows: list of (start, value); keep the launches of the last `steps` steps"""
    rows.sort()
    return [v for _, v in rows[-int(round(per_step * steps)):]]


# ---- durations per dispatch
dur = defaultdict(list)
for r in csv.DictReader(open(one(trace, "*kernel_trace.csv"))):
    dur[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))


def counters(d):
    acc = defaultdict(list)
    f = one(d, "*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    return acc


fs, ws = counters(fetch), counters(write)
# launches per step of every kernel: from the total count over the whole process (spin-up steps launch the same kernels
# per step, the ion passes excepted: those are counted over the tail of the run below)
table, traffic, ms_by_key = [], defaultdict(float), defaultdict(float)
for n, rows in dur.items():
    key = alias(n)
    if key is None:
        continue
    rows.sort()
    # launches per step in the timed region: launches between the start of the steps-th last launch of k_correct_all ...
    table.append((n, key, rows))
ref = sorted(dur[[n for n in dur if "k_correct_all<" in n or "k_sweep_x1<" in n or "k_sweep_x1_flat<" in n][0]])
t0 = ref[-steps][0]                       # start of the timed region's first hydro kernel of that name
# the ion step precedes the hydro kernels of a step: open the window at the last ion-step entry before t0
entries = sorted(s for n in dur if ("k_ion_pass<false, true" in n or "k_ion_begin" in n) for s, _ in dur[n] if s < t0)
if entries:
    t0 = entries[-1]
lines = []
tot_ms = 0.0
main_launches = {}                       # bench name -> (ms per step, launches per step) of its main kernel
for n, key, rows in table:
    sel = [v for s, v in rows if s >= t0]
    if not sel:
        continue
    f = [v for s, v in sorted(fs.get(n, []))][-len(sel):] if fs.get(n) else []
    w = [v for s, v in sorted(ws.get(n, []))][-len(sel):] if ws.get(n) else []
    ms = sum(sel) / len(sel)
    fk = sum(f) / len(f) if f else float("nan")
    wk = sum(w) / len(w) if w else float("nan")
    byts = (2.0 * fk + wk) * 1024.0
    per_step = len(sel) / steps
    tot_ms += ms * per_step
    ms_by_key[key] += ms * per_step
    if f and w:
        traffic[key] += byts * per_step          # per step; divided by launches per step of the bench name below
    if ms * per_step > main_launches.get(key, (0.0, 1.0))[0]:
        main_launches[key] = (ms * per_step, per_step)
    lines.append((ms * per_step, f"| `{n[:78]}` | {key} | {per_step:.2f} | {ms:.3f} | {ms * per_step:.2f} | {fk:.0f} | {wk:.0f} | {byts / zones:.0f} | "
                                 f"{byts / (ms * 1e-3) / 1e9 if ms > 0 else 0:.0f} |"))
with open(out_md, "w") as o:
    o.write(f"rocprofv3 --kernel-trace + --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps {steps}`; {workload}; "
            f"the last {steps} steps of the process = bench.py's timed region.  Sum of kernel time per step: {tot_ms:.2f} ms.\n\n")
    o.write("| kernel | bench name | launches / step | avg ms | ms / step | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM B/zone/launch (fetch x2 corr.) | eff. GB/s |\n")
    o.write("|---|---|---|---|---|---|---|---|---|\n")
    for _, ln in sorted(lines, reverse=True):
        o.write(ln + "\n")
    o.write("\nper bench name (ms / step): " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(ms_by_key.items(), key=lambda kv: -kv[1])) + "\n")
json.dump({"_comment": "HBM bytes per launch of bench.py's kernel names (ion_pass = one updating pass incl. its reduce / pick kernels) from rocprofv3 --pmc "
                       "FETCH_SIZE / WRITE_SIZE, separate passes, over the timed region; FETCH_SIZE doubled per the gfx950 correction of "
                       "MI355X_MICROARCH.md (HBM section); KiB -> bytes",
           "workload": workload, "source_fingerprint": _fp(), "commit": os.environ.get("AA_COMMIT"),
           "kernels": {k: v / main_launches[k][1] for k, v in sorted(traffic.items())}}, open(out_json, "w"), indent=1)
print(open(out_md).read())
