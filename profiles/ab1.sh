#!/bin/bash
# Same-box A/B of builds on the STATIONARY state of the headline window (165 spin-up steps, every build computes the same state: only for builds with right results).
# 512^3 ioniz_sphere, per-kernel hipEvent times.  usage: bash profiles/ab1.sh default v1 v2 ... [ROUNDS=2]
mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-2}); do
for v in "$@"; do
  if [ "$v" = default ]; then unset ATHENA_AMD_VARIANT; else export ATHENA_AMD_VARIANT=$v; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-burst --no-driver-window > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; continue; }
  python - "$v" <<'P'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_{v}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 1.0 and not a.startswith("ion")}
print(f"{v:10s}", k, "hydro", round(d["phases"]["hydro"]["ms_per_step"], 2), flush=True)
P
done; done 2>&1 | tee -a gpurun_out/ab1_log.txt
