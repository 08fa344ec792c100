#!/bin/bash
# k_bc by direction (the three launches of a bvals_mhd call differ in grid size): rocprofv3 kernel trace of a short 512^3 run
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/bc_trace
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/bc_trace -o t --output-format csv -- python3 $R/bench.py --spinup 0 --steps 4 --warmup 1 --no-burst --no-cpu-baseline --no-kernel-times > /dev/null 2> $R/gpurun_out/bc_trace.log
python3 - $(find $R/gpurun_out/bc_trace -name "*kernel_trace.csv" | head -1) <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_bc" in r["Kernel_Name"]:
        d[(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))*1e-3)
for k, v in sorted(d.items()):
    v.sort(); print("k_bc grid", k, "launches", len(v), "median us %.1f" % v[len(v)//2])
P
rm -rf $R/gpurun_out/bc_trace
