#!/bin/bash
# rocprofv3 kernel-trace summaries of the other measured configurations (profiles/r03_variants.txt holds their bench lines)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r03_variants_kernels.md
echo "rocprofv3 --kernel-trace --stats of \`bench.py <flags> --no-cpu-baseline\` (whole process: spin-up, warm-up and timed steps); top kernels by total time" > $OUT
run() { name=$1; shift
  rm -rf $R/gpurun_out/pw_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pw_$name -o t --output-format csv -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/pw_$name.json 2> $R/gpurun_out/pw_$name.log
  f=$(find $R/gpurun_out/pw_$name -name "*kernel_stats.csv" | head -1)
  echo "" >> $OUT; echo "### $name: bench.py $*" >> $OUT; echo "" >> $OUT
  python3 - "$f" >> $OUT <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | avg ms | total ms | % |"); print("|---|---|---|---|---|")
for r in rows[:10]:
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])*1e-6:.3f} | {float(r['TotalDurationNs'])*1e-6:.1f} | {float(r['Percentage']):.1f} |")
P
  rm -rf $R/gpurun_out/pw_$name
}
run blast --problem blast --steps 10 --warmup 2
run ifront256 --problem ifront --nx 256 --steps 10 --warmup 2
run vl --integrator vl --steps 10 --warmup 2 --spinup 30
run ppm --order 3 --steps 10 --warmup 2 --spinup 30
run smr --smr --steps 10 --warmup 2
cat $OUT
