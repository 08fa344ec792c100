#!/bin/bash
# A/B of the flat x1 sweep by its run-time switch (AA_X1_FLAT=0: a block per row piece), 512^3 burst window; run on the GPU box
for r in 1 2 3; do for v in 0 1; do
  AA_X1_FLAT=$v timeout -k 10 300 python bench.py --spinup burst --steps 6 --warmup 2 --no-cpu-baseline --no-burst > gpurun_out/x1f_$v.json 2> gpurun_out/x1f_$v.err || { echo "$v FAILED"; continue; }
  python - $v <<'P'
import json, sys
d = json.load(open(f"gpurun_out/x1f_{sys.argv[1]}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 1.0}
print("AA_X1_FLAT=" + sys.argv[1], k, "hydro", round(d["phases"]["hydro"]["ms_per_step"], 2), flush=True)
P
done; done
