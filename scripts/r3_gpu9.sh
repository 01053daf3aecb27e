set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/t9.log 2>&1 || { tail -60 gpurun_out/r3/t9.log; exit 1; }
tail -3 gpurun_out/r3/t9.log
MP_BENCH_E2E_VARIANTS="hostmerge:MP_HOST_MERGE=1" python3 bench.py --steps 3 --warmup 1 --no-pmc --legs e2e_cli --cpu-sample-pairs 20000 > gpurun_out/r3/bench_g.json 2> gpurun_out/r3/bench_g.err || { tail -30 gpurun_out/r3/bench_g.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_g.json'))
print(j['parity_sample']['records_equal'], j['parity_sample'].get('first_diff'))
for k,v in j['legs']['e2e_cli'].items():
    if isinstance(v, dict): print(k, v.get('pairs_per_s'), v.get('alignment_loop_s'), v.get('reader_ms_per_batch_steady'), v.get('stage_seconds'), v.get('error'))
"
