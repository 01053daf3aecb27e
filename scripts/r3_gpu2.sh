set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/t2.log 2>&1 || { tail -40 gpurun_out/r3/t2.log; exit 1; }
tail -3 gpurun_out/r3/t2.log
python3 bench.py --steps 5 --warmup 2 --no-pmc --legs e2e_cli > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err || { tail -20 gpurun_out/r3/bench_b.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_b.json'))
print(j['value'], j['ms_per_step'], j['kernel_ms_per_step'])
print(j['parity_sample']['records_equal'], j['cpu_baseline']['value'])
print(json.dumps(j['legs']['e2e_cli'], indent=1))
"
MP_TOUCH_AHEAD=0 python3 bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --legs e2e_cli > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err || { tail -20 gpurun_out/r3/bench_c.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_c.json'))
print('no touch-ahead:', json.dumps(j['legs']['e2e_cli']['plain'], indent=1))
"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
