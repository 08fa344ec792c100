#!/bin/bash
# A/B of the marching slope kernels (x2 / x3) by their run-time switch, 512^3 ioniz_sphere --order 3 (CTU + PPM)
for r in 1 2; do for v in 0 1; do
  AA_SLOPES_MARCH=$v timeout -k 10 400 python bench.py --order 3 --spinup 20 --steps 6 --warmup 2 --no-cpu-baseline --no-burst > gpurun_out/slm_$v.json 2> gpurun_out/slm_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/slm_$v.err; continue; }
  python - $v <<'P'
import json, sys
d = json.load(open(f"gpurun_out/slm_{sys.argv[1]}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 1.0 and a != "ion_pass"}
print("AA_SLOPES_MARCH=" + sys.argv[1], k, "hydro", round(d["phases"]["hydro"]["ms_per_step"], 2), flush=True)
P
done; done
