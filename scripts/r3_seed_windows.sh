# seed kernel: a start is ruled out by any of the 17-mers inside its first seed_min_length bases (six with soap4.ini).
# seeds equal with the pre-filter on and off / equal to the oracle's; a soak with soap4.ini drawn; the default step without legs.
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "prefilter or seed or human_genome_scale or soap4_ini" > gpurun_out/r3/t_seedwin.log 2>&1 || { tail -40 gpurun_out/r3/t_seedwin.log; exit 1; }
tail -2 gpurun_out/r3/t_seedwin.log
python -m pytest tests/test_gpu_soap4_cli.py -x -q -m gpu -k "soap4_ini" > gpurun_out/r3/t_seedwin2.log 2>&1 || { tail -40 gpurun_out/r3/t_seedwin2.log; exit 1; }
tail -2 gpurun_out/r3/t_seedwin2.log
timeout -k 10 400 python3 scripts/soak_parity.py --rounds 30 --seed 9002 --inis soap4.ini > gpurun_out/r3/soak_9002.log 2>&1 || { tail -5 gpurun_out/r3/soak_9002.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/r3/soak_9002.log
python3 bench.py --no-legs --no-pmc > gpurun_out/r3/bench_seedwin.json 2> gpurun_out/r3/bench_seedwin.err || { tail -30 gpurun_out/r3/bench_seedwin.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_seedwin.json'))
print(j['value'], j['ms_per_step'], j['single_context']['ms_per_step'], j['roofline']['frac'], j['roofline']['launch_ms'])
print(j['kernel_ms_per_step']); print(j['events_per_pair']); print(j['parity_sample']['records_equal'], j['parity_sample_nt2_ini']['records_equal'])
"
