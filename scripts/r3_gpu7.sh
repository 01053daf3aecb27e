set -e
mkdir -p gpurun_out/r3
t0=$(date +%s)
python3 bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || { tail -30 gpurun_out/r3/bench_default.err; exit 1; }
t1=$(date +%s)
echo "default bench: $((t1-t0)) s"
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_default.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('traffic_over_requested'), j['roofline'].get('frac_of_gather_ceiling'))
print(j['kernel_ms_per_step'])
print(j['parity_sample']['records_equal'], j['cpu_baseline']['value'])
for k,v in j['legs'].items(): print(k, v.get('pairs_per_s') if 'pairs_per_s' in v else {kk:(vv.get('pairs_per_s') if isinstance(vv,dict) else None) for kk,vv in v.items()}, v.get('error'))
"
bash scripts/collect_profiles.sh r03 > gpurun_out/r3/collect_r03.log 2>&1 || { tail -30 gpurun_out/r3/collect_r03.log; exit 1; }
tail -3 gpurun_out/r3/collect_r03.log
ls gpurun_out/prof_r03/summary
