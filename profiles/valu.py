#!/usr/bin/env python3
"""VALU utilisation per kernel from one rocprofv3 --pmc pass:
  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ counters are summed over the 8 XCDs, GRBM_GUI_ACTIVE as well (8 x the kernel's cycles): VALU busy =
SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
usage: valu.py <pmc_dir> <zones> > profiles/rNN_valu.md"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
zones = float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | launches | kernel cycles (M) | VALU busy % | wave-level VALU instr / launch (G) | per zone (lane instr) | waves parked (s_waitcnt, barrier) % | issue stall % |")
print("|---|---|---|---|---|---|---|---|")
tot = 0.0
rows = []
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc < 2e4:
        continue
    wc = m["SQ_WAVE_CYCLES"]
    rows.append((cyc * len(c["GRBM_GUI_ACTIVE"]), n, len(c["GRBM_GUI_ACTIVE"]), cyc, 100 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                 m["SQ_INSTS_VALU"], 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc))
for _, n, calls, cyc, busy, insts, w1, w2 in sorted(rows, reverse=True):
    short = re.sub(r"\(.*", "", n)[:64]
    print(f"| `{short}` | {calls} | {cyc / 1e6:.2f} | {busy:.0f} | {insts / 1e9:.2f} | {insts * 64 / zones:.0f} | {w1:.0f} | {w2:.0f} |")
