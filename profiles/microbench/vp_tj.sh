for r in 1 2; do for v in default vp4; do
  if [ "$v" = default ]; then unset ATHENA_AMD_VARIANT; else export ATHENA_AMD_VARIANT=$v; fi
  timeout -k 10 400 python bench.py --integrator vl --spinup 20 --steps 6 --warmup 2 --no-cpu-baseline --no-burst > gpurun_out/vp_$v.json 2> gpurun_out/vp_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/vp_$v.err; continue; }
  python - $v <<'P'
import json, sys
d = json.load(open(f"gpurun_out/vp_{sys.argv[1]}.json"))
k = {a: round(b, 2) for a, b in d["kernel_ms_per_step"].items() if b > 0.5 and a != "ion_pass"}
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 2), k, flush=True)
P
done; done
