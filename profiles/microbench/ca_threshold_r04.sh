#!/bin/bash
# k_correct_all (with the x1 and x3 first passes on board) against the tile kernels on small Grids: where the library should switch.
# stationary window of the headline deck / blast, no per-stage events.  Run on the GPU box from the repo root.
one() { p=$1; n=$2; ca=$3; AA_CORRECT_ALL=$ca timeout -k 10 200 python bench.py --problem $p --nx $n --steps 40 --warmup 5 --no-cpu-baseline --no-burst --no-driver-window --no-kernel-times 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$p nx $n correct_all=$ca ms/step %.4f value %.3e' % (d['ms_per_step'], d['value']), flush=True)"; }
for n in 64 72 80 96 112; do for ca in 0 1; do one ioniz_sphere $n $ca; done; done
for n in 64 72 80 96; do for ca in 0 1; do one blast $n $ca; done; done
for ca in 0 1 default; do if [ $ca = default ]; then unset AA_CORRECT_ALL; else export AA_CORRECT_ALL=$ca; fi; timeout -k 10 200 python bench.py --smr --smr-deck --steps 40 --warmup 10 --no-cpu-baseline --no-burst --no-driver-window --no-kernel-times 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('deck2 correct_all=$ca ms/step %.4f value %.3e' % (d['ms_per_step'], d['value']), flush=True)"; done
