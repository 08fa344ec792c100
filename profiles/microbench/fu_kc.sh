# chunk length (planes per block) of k_flux2_update against the size of the Grid: AA_FU_KC overrides the default rule
for spec in "${@:-128:4,8,16,32 192:8,16,32,64 256:8,16,32,64 384:16,32,64 512:16,32,64,128}"; do
  n=${spec%%:*}; for kc in default $(echo ${spec#*:} | tr , ' '); do
  if [ $kc = default ]; then unset AA_FU_KC; else export AA_FU_KC=$kc; fi
  timeout -k 10 300 python bench.py --nx $n --steps 20 --warmup 3 --no-cpu-baseline --no-burst 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']
print('nx $n fu_kc $kc: ms/step %.3f hydro %.3f correct_all %.3f flux2_update %.3f' % (d['ms_per_step'], d['phases']['hydro']['ms_per_step'], k.get('correct_all',0), k.get('flux2_update',0)), flush=True)"
done; done
