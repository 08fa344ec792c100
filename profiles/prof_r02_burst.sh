#!/bin/bash
# the rocprofv3 passes behind profiles/r02_burst_*: bench.py exactly as the driver runs it (--steps 20 --warmup 5)
set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
ARGS="--spinup 19 --steps 6 --warmup 0 --no-cpu-baseline"
rm -rf $R/gpurun_out/r02_burst_trace $R/gpurun_out/r02_burst_fetch $R/gpurun_out/r02_burst_write $R/gpurun_out/r02_burst_sq
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_burst_trace -o t --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/r02_burst_bench_under_rocprof.json 2> $R/gpurun_out/r02_burst_trace.log
echo trace done
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r02_burst_fetch -o f --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_burst_fetch.log
echo fetch done
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r02_burst_write -o w --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_burst_write.log
echo write done
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r02_burst_sq -o sq --output-format csv -- python3 $R/bench.py $ARGS --no-kernel-times > /dev/null 2> $R/gpurun_out/r02_burst_sq.log
echo sq done
cd $R
python3 profiles/summarize_r02.py gpurun_out/r02_burst_trace gpurun_out/r02_burst_fetch gpurun_out/r02_burst_write 134217728 6 "ioniz_sphere 512x512x512, burst regime (--spinup 19)" gpurun_out/r02_burst_kernels.md gpurun_out/r02_burst_traffic.json > /dev/null
python3 profiles/valu.py gpurun_out/r02_burst_sq 134217728 > gpurun_out/r02_burst_valu.md
cp `find gpurun_out/r02_burst_trace -name "*kernel_stats.csv" | head -1` gpurun_out/r02_burst_kernel_stats.csv
# keep the merged-back payload small: the per-dispatch CSVs stay on the box
rm -rf gpurun_out/r02_burst_trace gpurun_out/r02_burst_fetch gpurun_out/r02_burst_write gpurun_out/r02_burst_sq
cat gpurun_out/r02_burst_kernels.md
