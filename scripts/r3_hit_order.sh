# pair hits of a pair leave the device in candidate order (not in the order wavefronts stored them): the whole GPU suite twice, a soak, the default step
set -e
mkdir -p gpurun_out/r3
for k in 1 2; do python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_order_$k.log 2>&1 || { tail -30 gpurun_out/r3/t_order_$k.log; exit 1; }; tail -1 gpurun_out/r3/t_order_$k.log; done
timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed 9301 > gpurun_out/r3/soak_order.log 2>&1 || { tail -5 gpurun_out/r3/soak_order.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/r3/soak_order.log
python3 bench.py --no-legs --no-pmc > gpurun_out/r3/bench_order.json 2> gpurun_out/r3/bench_order.err || { tail -30 gpurun_out/r3/bench_order.err; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r3/bench_order.json'))
print(j['value'], j['ms_per_step'], j['single_context']['ms_per_step']); print(j['kernel_ms_per_step']); print(j['parity_sample']['records_equal'], j['parity_sample_nt2_ini']['records_equal'])"
