#!/bin/bash
# like ab.sh, on the hydro-only 512^3 blast (NVAR = 5, no gravity: other template instantiations of the same kernels)
for r in $(seq 1 ${ROUNDS:-2}); do
for v in "$@"; do
  if [ "$v" = default ]; then unset ATHENA_AMD_VARIANT; else export ATHENA_AMD_VARIANT=$v; fi
  timeout -k 10 300 python bench.py --problem blast --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('blast %-12s' % '$v', {a: round(b, 2) for a, b in d['kernel_ms_per_step'].items() if b > 1.0}, 'ms/step', round(d['ms_per_step'], 2), flush=True)"
done; done 2>&1 | tee -a gpurun_out/ab_log.txt
