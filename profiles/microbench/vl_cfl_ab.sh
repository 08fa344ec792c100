#!/bin/bash
# van Leer integrator at 512^3: new_dt's maxima from k_update (AA_CFL_FUSED default) against k_cfl's own sweep (AA_CFL_FUSED=0)
for r in 1 2; do for v in 0 1; do
  AA_CFL_FUSED=$v timeout -k 10 400 python bench.py --integrator vl --spinup 20 --steps 6 --warmup 2 --no-cpu-baseline --no-burst > gpurun_out/vlc_$v.json 2> gpurun_out/vlc_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/vlc_$v.err; continue; }
  python - $v <<'P'
import json, sys
d = json.load(open(f"gpurun_out/vlc_{sys.argv[1]}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 0.5 and a != "ion_pass"}
print("AA_CFL_FUSED=" + sys.argv[1], "ms/step", round(d["ms_per_step"], 2), k, flush=True)
P
done; done
