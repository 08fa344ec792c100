for n in 128 192 256 512; do for v in default noedge default noedge; do
  if [ "$v" = default ]; then unset ATHENA_AMD_VARIANT; else export ATHENA_AMD_VARIANT=$v; fi
  timeout -k 10 300 python bench.py --nx $n --spinup 0 --steps 10 --warmup 3 --no-cpu-baseline --no-burst --no-driver-window 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print('$n $v', 'ms/step %.4f' % d['ms_per_step'], 'correct_all %.4f' % k.get('correct_all', 0), 'flux2_update %.4f' % k.get('flux2_update',0), flush=True)"
done; done
