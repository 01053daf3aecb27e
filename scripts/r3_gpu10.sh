set -e
mkdir -p gpurun_out/r3
MP_BENCH_E2E_VARIANTS="r8p8:MP_READER_THREADS=8,MP_PRINTER_THREADS=8;r4p8:MP_READER_THREADS=4,MP_PRINTER_THREADS=8;w4:MP_WORKERS_PER_GPU=4" python3 bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --legs e2e_cli > gpurun_out/r3/bench_h.json 2> gpurun_out/r3/bench_h.err || { tail -30 gpurun_out/r3/bench_h.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_h.json'))
for k,v in j['legs']['e2e_cli'].items():
    if isinstance(v, dict) and 'pairs_per_s' in v: print(k, v.get('pairs_per_s'), v.get('alignment_loop_s'), v.get('reader_ms_per_batch_steady'), v.get('stage_seconds'), v.get('error'))
"
