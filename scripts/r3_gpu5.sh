set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu -k "cli_stdout or paired_only" > gpurun_out/r3/t5.log 2>&1 || { tail -40 gpurun_out/r3/t5.log; exit 1; }
tail -2 gpurun_out/r3/t5.log
MP_BENCH_E2E_VARIANTS="r6p8:MP_READER_THREADS=6,MP_PRINTER_THREADS=8;r8p8:MP_READER_THREADS=8,MP_PRINTER_THREADS=8;r3p10:MP_READER_THREADS=3,MP_PRINTER_THREADS=10" python3 bench.py --steps 3 --warmup 1 --no-pmc --legs e2e_cli,nt_replica --cpu-sample-pairs 20000 > gpurun_out/r3/bench_f.json 2> gpurun_out/r3/bench_f.err || { tail -30 gpurun_out/r3/bench_f.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_f.json'))
for k,v in j['legs']['e2e_cli'].items():
    if isinstance(v, dict): print(k, v.get('pairs_per_s'), v.get('alignment_loop_s'), v.get('reader_ms_per_batch_steady'), v.get('stage_seconds'))
n=j['legs']['nt_replica']
print(n.get('pairs_per_s'), n.get('cpu_baseline'), n.get('error'))
"
