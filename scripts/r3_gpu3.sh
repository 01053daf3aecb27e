set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu -k "nine or resident or chain" > gpurun_out/r3/t3.log 2>&1 || { tail -40 gpurun_out/r3/t3.log; exit 1; }
tail -3 gpurun_out/r3/t3.log
MP_BENCH_E2E_VARIANTS="r8p8:MP_READER_THREADS=8,MP_PRINTER_THREADS=8;r4p16:MP_READER_THREADS=4,MP_PRINTER_THREADS=16;r2p8:MP_READER_THREADS=2,MP_PRINTER_THREADS=8;w2:MP_WORKERS_PER_GPU=2;w4:MP_WORKERS_PER_GPU=4;T32:MP_READER_THREADS=8,MP_PRINTER_THREADS=16" python3 bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --legs e2e_cli > gpurun_out/r3/bench_d.json 2> gpurun_out/r3/bench_d.err || { tail -20 gpurun_out/r3/bench_d.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_d.json'))
for k,v in j['legs']['e2e_cli'].items():
    if isinstance(v, dict): print(k, v.get('pairs_per_s'), v.get('alignment_loop_s'), v.get('stage_seconds'))
"
