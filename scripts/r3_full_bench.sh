set -e
mkdir -p gpurun_out/r3
t0=$(date +%s)
python3 bench.py > gpurun_out/r3/bench_default2.json 2> gpurun_out/r3/bench_default2.err || { tail -30 gpurun_out/r3/bench_default2.err; exit 1; }
t1=$(date +%s)
echo "default bench: $((t1-t0)) s"
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_default2.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('traffic_over_requested'), j['roofline'].get('frac_of_gather_ceiling'))
print(j['parity_sample']['records_equal'], j['cpu_baseline']['value'])
for k,v in j['legs'].items():
    if k=='e2e_cli':
        for kk,vv in v.items():
            if isinstance(vv,dict) and 'pairs_per_s' in vv: print('  e2e', kk, vv['pairs_per_s'])
        print('  e2e resident', {a:(b.get('pairs_per_s'), b.get('stage_seconds','')[-60:]) for a,b in v.get('resident_chunks',{}).items() if isinstance(b,dict)})
    else: print(k, v.get('pairs_per_s'), v.get('error'))
"
