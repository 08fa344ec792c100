#!/bin/bash
# the rocprofv3 passes behind profiles/r02_*: bench.py exactly as the driver runs it (--steps 20 --warmup 5).
# Run from the repo root on the GPU box: gpurun -- bash profiles/prof_r02.sh ; outputs land in gpurun_out/ and are copied here.
set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
rm -rf $R/gpurun_out/r02_trace $R/gpurun_out/r02_fetch $R/gpurun_out/r02_write $R/gpurun_out/r02_sq
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_trace -o t --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/r02_trace.log
echo trace done
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r02_fetch -o f --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_fetch.log
echo fetch done
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r02_write -o w --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_write.log
echo write done
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r02_sq -o sq --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_sq.log
echo sq done
cd $R
python3 profiles/summarize_r02.py gpurun_out/r02_trace gpurun_out/r02_fetch gpurun_out/r02_write 134217728 20 "ioniz_sphere 512x512x512" gpurun_out/r02_kernels.md gpurun_out/r02_traffic.json > /dev/null
python3 profiles/valu.py gpurun_out/r02_sq 134217728 > gpurun_out/r02_valu.md
cp `find gpurun_out/r02_trace -name "*kernel_stats.csv" | head -1` gpurun_out/r02_kernel_stats.csv
# keep the merged-back payload small: the per-dispatch CSVs stay on the box
rm -rf gpurun_out/r02_trace gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_sq
cat gpurun_out/r02_kernels.md
