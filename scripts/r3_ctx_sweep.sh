# contexts per GPU on the default step after the round's kernel changes (seeding 7.5 -> 3.5 ms, deep-DP fill 11.6 -> 10.0 ms)
set -e
mkdir -p gpurun_out/r3
for n in 2 3 4 6; do
  python3 bench.py --no-legs --no-pmc --no-cpu-baseline --steps 6 --contexts $n > gpurun_out/r3/ctx_$n.json 2> gpurun_out/r3/ctx.err
  python3 -c "
import json; j=json.load(open('gpurun_out/r3/ctx_$n.json')); print('contexts $n:', round(j['value']/1e6,2), 'M pairs/s,', round(j['ms_per_step'],2), 'ms per step')"
done
