#!/bin/bash
# planes per block of k_correct_all / k_flux2_update at 128^3 and 192^3 with the round-4 kernels (AA_CA_KC, AA_FU_KC), no per-stage events
for n in 128 192; do for kc in 0 4 8 16 32; do
AA_CA_KC=$kc timeout -k 10 200 python bench.py --nx $n --steps 40 --warmup 5 --no-cpu-baseline --no-burst --no-driver-window --no-kernel-times 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('nx $n AA_CA_KC=$kc ms/step %.4f value %.3e' % (d['ms_per_step'], d['value']), flush=True)"
done; for kc in 4 8 16; do
AA_FU_KC=$kc timeout -k 10 200 python bench.py --nx $n --steps 40 --warmup 5 --no-cpu-baseline --no-burst --no-driver-window --no-kernel-times 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('nx $n AA_FU_KC=$kc ms/step %.4f value %.3e' % (d['ms_per_step'], d['value']), flush=True)"
done; done
