# the seeding kernel's build variants on the default step after the six-window pre-filter rule: one item per lane (default), two (MP_SEED_ITEMS=2),
# six waves per SIMD (MP_SEED_OCC=6)
set -e
mkdir -p gpurun_out/r3
for v in "base" "MP_SEED_ITEMS=2" "MP_SEED_OCC=6"; do
  if [ "$v" = base ]; then python3 bench.py --no-legs --no-pmc --no-cpu-baseline --steps 5 > gpurun_out/r3/seedvar_$v.json 2> gpurun_out/r3/seedvar.err
  else env $v python3 bench.py --no-legs --no-pmc --no-cpu-baseline --steps 5 > gpurun_out/r3/seedvar_$v.json 2> gpurun_out/r3/seedvar.err; fi
  python3 -c "
import json,sys; j=json.load(open('gpurun_out/r3/seedvar_$v.json'))
print('$v', round(j['value']/1e6,2), 'M pairs/s; seed', round(j['kernel_ms_per_step']['seed'],3), 'ms; one context', round(j['single_context']['ms_per_step'],2))"
done
