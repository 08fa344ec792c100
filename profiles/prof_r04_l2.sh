#!/bin/bash
# L2 (TCC) counters of the hydro chain's kernels, 512^3 ioniz_sphere stationary window: requests, hits, misses and the memory-side
# read / write requests per kernel (where does the fetch beyond the operands come from).
# gpurun -- bash profiles/prof_r04_l2.sh ; the per-kernel table lands in gpurun_out/r04_l2.md
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
ARGS="--steps 4 --warmup 1 --no-burst --no-cpu-baseline --no-kernel-times"
rocprofv3 -L > $O/r04_counters_list.txt 2>&1
rm -rf $O/r04_l2a $O/r04_l2b
timeout -k 10 400 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum -d $O/r04_l2a -o a --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $O/r04_l2a.log || { echo "pass a failed"; tail -5 $O/r04_l2a.log; }
echo pass a done
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/r04_l2b -o b --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $O/r04_l2b.log || { echo "pass b failed"; tail -5 $O/r04_l2b.log; }
echo pass b done
cd $R
python3 profiles/l2_table.py gpurun_out/r04_l2a gpurun_out/r04_l2b 134217728 > gpurun_out/r04_l2.md
rm -rf gpurun_out/r04_l2a gpurun_out/r04_l2b
cat gpurun_out/r04_l2.md
