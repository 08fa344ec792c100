#!/bin/bash
# Clocks and socket power while bench.py's stationary window runs: is the step clock-limited by the power cap?
# usage (GPU box, repo root): bash profiles/microbench/power_trace.sh
python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-burst --no-cpu-baseline > gpurun_out/pt_bench.json 2> gpurun_out/pt_bench.err &
BP=$!
for i in $(seq 1 80); do
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
  echo "t=$i $(rocm-smi -d 0 -P -c -t --json 2>/dev/null | tr -d '\n' | cut -c1-600)"
done > gpurun_out/pt_samples.txt
wait $BP
rocm-smi -d 0 --showmaxpower --showpowercap 2>/dev/null | grep -i "power\|cap" | head -5
tail -c 300 gpurun_out/pt_bench.json | head -c 0; python3 -c "
import json; d=json.loads(open('gpurun_out/pt_bench.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['kernel_ms_per_step'])"
