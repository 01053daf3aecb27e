# merge_kernel: four reads per wavefront, one 16-lane row each (MP_MERGE_ROWS=0: every read by a whole wavefront)
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_soap4_cli.py -x -q -m gpu > gpurun_out/r3/t_rows.log 2>&1 || { tail -30 gpurun_out/r3/t_rows.log; exit 1; }
tail -1 gpurun_out/r3/t_rows.log
timeout -k 10 400 python3 scripts/soak_parity.py --rounds 30 --seed ${SOAK_SEED:-9502} > gpurun_out/r3/soak_rows.log 2>&1 || { tail -5 gpurun_out/r3/soak_rows.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/r3/soak_rows.log
for v in 0 1; do
  MP_MERGE_ROWS=$v python3 bench.py --no-legs --no-pmc --steps 6 > gpurun_out/r3/bench_rows_$v.json 2> gpurun_out/r3/bench_rows.err || { tail -20 gpurun_out/r3/bench_rows.err; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_rows_$v.json'))
print('MP_MERGE_ROWS=$v', round(j['value']/1e6,2), 'one context', round(j['single_context']['ms_per_step'],2), 'merge', round(j['kernel_ms_per_step']['merge'],3), j['parity_sample']['records_equal'], j['parity_sample_nt2_ini']['records_equal'])"
done
