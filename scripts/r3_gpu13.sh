set -e
mkdir -p gpurun_out/r3
python3 bench.py --steps 3 --warmup 1 --no-pmc --no-legs --cpu-sample-pairs 40000 > gpurun_out/r3/bench_p.json 2> gpurun_out/r3/bench_p.err || { tail -30 gpurun_out/r3/bench_p.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_p.json'))
print(j['parity_sample'])
print(j.get('parity_sample_nt2_ini'))
"
timeout -k 10 700 python3 scripts/soak_parity.py --rounds 60 --seed 77 > gpurun_out/r3/soak_77.log 2>&1 || { tail -5 gpurun_out/r3/soak_77.log | cut -c1-500; exit 1; }
tail -1 gpurun_out/r3/soak_77.log
