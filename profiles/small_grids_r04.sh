#!/bin/bash
# The reference's own deck sizes on one GPU: 80^3 / 128^3 single level, the first two levels of
# tst/massloss/athinput.ioniz_sphere_hires (80^3 + 52^3) and all five.  One line per case + the kernel table of the timed region.
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-burst --no-driver-window > gpurun_out/sg_$name.json 2> gpurun_out/sg_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/sg_$name.err; return; }
python - "$name" <<'P'
import json, sys
name = sys.argv[1]
d = json.load(open(f"gpurun_out/sg_{name}.json"))
k = d.get("kernel_ms_per_step") or d.get("kernel_ms_per_step_rank0") or {}
n = d.get("kernel_launches_per_step", {})
nsub = d["config"].get("radiation_subcycles_per_step", d["config"].get("subcycle_trace_per_level", [None])[-1])
print(f"{name:10s} ms/step {d['ms_per_step']:.4f}  value {d['value']:.3e}  nsub {nsub}  host_syncs/step {d.get('host_syncs_per_step')}  unattributed {d.get('phases', {}).get('unattributed_ms')}")
if k: print("   " + "  ".join(f"{a} {b*1e3:.0f}us" + (f"/{n[a]:.0f}" if a in n else "") for a, b in list(k.items())[:16]), flush=True)
P
}
run nx80 --nx 80 --steps 40 --warmup 5
run nx128nk --nx 128 --steps 40 --warmup 5 --no-kernel-times
run deck2nk --smr --smr-deck --steps 40 --warmup 10 --no-kernel-times
run nx128 --nx 128 --steps 40 --warmup 5
run nx192 --nx 192 --steps 20 --warmup 5
run deck2 --smr --smr-deck --steps 40 --warmup 10
AA_SMR_DEEP_RADIATION=fixed run deck5 --smr --smr-deck --smr-levels 5 --steps 10 --warmup 10
