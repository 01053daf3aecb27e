set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu -k "cli" > gpurun_out/r3/t4.log 2>&1 || { tail -40 gpurun_out/r3/t4.log; exit 1; }
tail -3 gpurun_out/r3/t4.log
python3 bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --legs e2e_cli,nt_replica > gpurun_out/r3/bench_e.json 2> gpurun_out/r3/bench_e.err || { tail -30 gpurun_out/r3/bench_e.err; exit 1; }
grep "NT replica" gpurun_out/r3/bench_e.err || true
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_e.json'))
for k,v in j['legs']['e2e_cli'].items():
    if isinstance(v, dict): print(k, v.get('pairs_per_s'), v.get('alignment_loop_s'), v.get('stage_seconds'))
print(json.dumps(j['legs']['nt_replica'], indent=1))
"
