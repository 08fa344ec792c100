#!/bin/bash
# cell-updates/s of the headline deck against the size of the cube on one GPU (stationary window, no CPU leg, no burst window)
for n in 128 192 256 384 512 640; do
  timeout -k 10 500 python bench.py --nx $n --steps 10 --warmup 2 --no-cpu-baseline --no-burst --no-driver-window 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']
print('nx %4d  ms/step %8.3f  cell-updates/s %.3e  nsub %s  hydro %.3f ms  ns/zone %.4f  frac(96 B) %.4f' % ($n, d['ms_per_step'], d['value'], d['config'].get('radiation_subcycles_per_step'), d['phases']['hydro']['ms_per_step'], d['phases']['hydro']['ns_per_zone'], d['roofline']['frac']), flush=True)"
done
