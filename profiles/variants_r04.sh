#!/bin/bash
# the other measured variants (one line each), round 4; run from the repo root on the GPU box
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-burst --no-driver-window > gpurun_out/var_$name.json 2> gpurun_out/var_$name.err; python - <<P
import json
d=json.load(open("gpurun_out/var_$name.json"))
ph=d.get("phases",{})
print("$name", "ms/step", round(d["ms_per_step"],2), "value %.3e" % d["value"], "nsub", d["config"].get("radiation_subcycles_per_step"), "hydro", round(ph.get("hydro",{}).get("ms_per_step",0),2), "sub", ph.get("subcycle",{}).get("ms_per_subcycle"), "dom", d.get("roofline",{}).get("kernel"), round(d.get("roofline",{}).get("frac",0),3))
P
}
run strict --strict --steps 10 --warmup 2
run blast --problem blast --steps 10 --warmup 2
run ifront256 --problem ifront --nx 256 --steps 10 --warmup 2
run vl --integrator vl --steps 10 --warmup 2
run ppm --order 3 --steps 10 --warmup 2
run slab --ionized-slab --spinup 0 --steps 5 --warmup 2
run smr --smr --steps 10 --warmup 2
run inlib2 --inlib 2 --steps 10 --warmup 2
