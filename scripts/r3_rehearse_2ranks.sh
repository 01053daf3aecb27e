set -e
mkdir -p gpurun_out/r3
echo skip-tests

env -u RANK -u LOCAL_RANK -u WORLD_SIZE MP_BENCH_SHARE_GPU=1 MP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-legs --no-pmc --no-cpu-baseline --genome-mbp 400 > gpurun_out/r3/bench_gpus2.json 2> gpurun_out/r3/bench_gpus2.err || { tail -20 gpurun_out/r3/bench_gpus2.err; exit 1; }
python3 -c "import json; j=json.loads([l for l in open('gpurun_out/r3/bench_gpus2.json') if l.startswith('{')][0]); print('gpus2:', j['n_gpus'], j['value'], j['hit_count_allreduce'])"
python3 bench.py --steps 5 --warmup 2 --no-pmc > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err || { tail -20 gpurun_out/r3/bench_a.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_a.json'))
print(j['value'], j['ms_per_step'], j['config']['ini'], j['kernel_ms_per_step'])
print(j['parity_sample'])
print({k:(v.get('pairs_per_s'),v.get('kernel_ms_per_step')) for k,v in j['legs'].items()})
"
