# Row N1 on the one-GPU box: `python3 bench.py --gpus 2` with NO launcher and a clean environment must become two ranks by itself
# (both on device 0, gloo instead of RCCL: MP_BENCH_SHARE_GPU / MP_BENCH_BACKEND are rehearsal knobs).  Run through gpurun from the repo root.
set -e
mkdir -p gpurun_out/r3
env -u RANK -u LOCAL_RANK -u WORLD_SIZE MP_BENCH_SHARE_GPU=1 MP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-legs --no-pmc --no-cpu-baseline --genome-mbp 400 > gpurun_out/r3/bench_gpus2.json 2> gpurun_out/r3/bench_gpus2.err || { tail -20 gpurun_out/r3/bench_gpus2.err; exit 1; }
python3 -c "import json; j=json.loads([l for l in open('gpurun_out/r3/bench_gpus2.json') if l.startswith('{')][0]); print('gpus2:', j['n_gpus'], j['value'], j['hit_count_allreduce'])"
