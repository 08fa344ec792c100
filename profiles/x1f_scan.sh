#!/bin/bash
# k_correct_all with / without the x1 first pass on board (CA_X1F) over the other workloads, same box.  usage: bash profiles/x1f_scan.sh default x1foff ...
mkdir -p gpurun_out
run() { v=$1; name=$2; shift 2
  if [ "$v" = default ]; then unset ATHENA_AMD_VARIANT; else export ATHENA_AMD_VARIANT=$v; fi
  timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-burst --no-driver-window > gpurun_out/xs_${v}_$name.json 2> gpurun_out/xs_${v}_$name.err || { echo "$v $name FAILED"; tail -3 gpurun_out/xs_${v}_$name.err; return; }
  python - "$v" "$name" <<'P'
import json, sys
v, name = sys.argv[1:3]
d = json.load(open(f"gpurun_out/xs_{v}_{name}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 0.5 and not a.startswith("ion")}
print(f"{name:10s} {v:8s}", k, "hydro", round(d["phases"]["hydro"]["ms_per_step"], 2), "ms/step", round(d["ms_per_step"], 2), flush=True)
P
}
for r in 1 2; do for v in "$@"; do
  run $v blast --problem blast --steps 10 --warmup 2
  run $v ifront256 --problem ifront --nx 256 --steps 10 --warmup 2
  run $v ppm --order 3 --steps 10 --warmup 2
  run $v nx256 --nx 256 --steps 20 --warmup 3
done; done 2>&1 | tee -a gpurun_out/x1f_scan.txt
for v in "$@"; do [ "$v" = default ] && unset ATHENA_AMD_VARIANT && run default strict --strict --steps 10 --warmup 2; done | tee -a gpurun_out/x1f_scan.txt
