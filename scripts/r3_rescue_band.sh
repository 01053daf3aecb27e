# rescue rounds: the instances that pass the filter go through the band scheme (windows of read + 60) instead of the full table
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_soap4_cli.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3/t_resband.log 2>&1 || { tail -40 gpurun_out/r3/t_resband.log; exit 1; }
tail -2 gpurun_out/r3/t_resband.log
timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed ${SOAK_SEED:-9201} > gpurun_out/r3/soak_resband.log 2>&1 || { tail -5 gpurun_out/r3/soak_resband.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/r3/soak_resband.log
python3 bench.py --no-pmc --legs nt2_ini_same_reads,C3_divergent,C4_2x250,C5_mixed_rescue > gpurun_out/r3/bench_resband.json 2> gpurun_out/r3/bench_resband.err || { tail -30 gpurun_out/r3/bench_resband.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_resband.json'))
print(j['value'], j['ms_per_step'], j['single_context']['ms_per_step'])
print(j['kernel_ms_per_step']); print(j['parity_sample']['records_equal'], j['parity_sample_nt2_ini']['records_equal'])
for k,v in j['legs'].items(): print(k, v.get('pairs_per_s'), v.get('kernel_ms_per_step'))
"
