#!/bin/bash
# chunk length (planes per block) of k_correct_all at 512^3: AA_CA_KC overrides the default (64 with the x3 first pass inside)
for r in 1 2; do for kc in ${@:-64 86 103 128 172}; do
  AA_CA_KC=$kc timeout -k 10 300 python bench.py --spinup burst --steps 6 --warmup 2 --no-cpu-baseline --no-burst 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']
print('ca_kc $kc: hydro %.3f correct_all %.3f flux2_update %.3f sweep_x1 %.3f sweep_x2 %.3f' % (d['phases']['hydro']['ms_per_step'], k.get('correct_all',0), k.get('flux2_update',0), k.get('sweep_x1',0), k.get('sweep_x2',0)), flush=True)"
done; done
