#!/bin/bash
# The rocprofv3 passes behind profiles/r04_*: bench.py as the driver runs it (--steps 20 --warmup 5), one window per call:
#   prof_r04.sh stationary   the headline window (--no-burst --no-cpu-baseline: the same timed region without the second
#                            window and the CPU leg behind it)
#   prof_r04.sh burst        the second window on its own (--spinup burst --steps 6 --warmup 0)
# Separate passes: kernel trace; FETCH_SIZE; WRITE_SIZE; SQ counters (MI355X_MICROARCH.md, HBM / rocprofv3 section).
# Run from the repo root on the GPU box: gpurun -- 'AA_COMMIT=<git rev-parse --short HEAD> bash profiles/prof_r04.sh stationary' ; the summaries land in gpurun_out/
# and are copied to profiles/.
set -e
W=${1:-stationary}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
if [ "$W" = burst ]; then ARGS="--spinup burst --steps 6 --warmup 0 --no-burst --no-cpu-baseline --no-driver-window"; STEPS=6; P=r04_burst; WL="ioniz_sphere 512x512x512, burst regime (--spinup burst)"
else ARGS="--steps 20 --warmup 5 --no-burst --no-cpu-baseline --no-driver-window"; STEPS=20; P=r04; WL="ioniz_sphere 512x512x512"; fi
O=$R/gpurun_out
rm -rf $O/${P}_trace $O/${P}_fetch $O/${P}_write $O/${P}_sq
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/${P}_trace -o t --output-format csv -- python3 $R/bench.py $ARGS > $O/${P}_bench_under_rocprof.json 2> $O/${P}_trace.log
echo trace done
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $O/${P}_fetch -o f --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $O/${P}_fetch.log
echo fetch done
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $O/${P}_write -o w --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $O/${P}_write.log
echo write done
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/${P}_sq -o sq --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $O/${P}_sq.log
echo sq done
cd $R
python3 profiles/summarize_r02.py gpurun_out/${P}_trace gpurun_out/${P}_fetch gpurun_out/${P}_write 134217728 $STEPS "$WL" gpurun_out/${P}_kernels.md gpurun_out/${P}_traffic.json > /dev/null
python3 profiles/valu.py gpurun_out/${P}_sq 134217728 > gpurun_out/${P}_valu.md
cp `find gpurun_out/${P}_trace -name "*kernel_stats.csv" | head -1` gpurun_out/${P}_kernel_stats.csv
# keep the merged-back payload small: the per-dispatch CSVs stay on the box
rm -rf gpurun_out/${P}_trace gpurun_out/${P}_fetch gpurun_out/${P}_write gpurun_out/${P}_sq
cat gpurun_out/${P}_kernels.md
