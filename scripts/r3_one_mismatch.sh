# deep DP: one-mismatch instances on a single matching diagonal finished by the pre-filter (no table): band tests, a soak, the default step
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "one_mismatch or banded or plain_diagonal or human_genome_scale" > gpurun_out/r3/t_onemm.log 2>&1 || { tail -40 gpurun_out/r3/t_onemm.log; exit 1; }
tail -2 gpurun_out/r3/t_onemm.log
timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed ${SOAK_SEED:-9103} > gpurun_out/r3/soak_onemm.log 2>&1 || { tail -5 gpurun_out/r3/soak_onemm.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/r3/soak_onemm.log
python3 bench.py --no-legs --no-pmc > gpurun_out/r3/bench_onemm.json 2> gpurun_out/r3/bench_onemm.err || { tail -30 gpurun_out/r3/bench_onemm.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_onemm.json'))
print(j['value'], j['ms_per_step'], j['single_context']['ms_per_step'])
print(j['kernel_ms_per_step']); print(j['dp']['cells_computed_per_step']); print(j['parity_sample']['records_equal'], j['parity_sample_nt2_ini']['records_equal'])
"
