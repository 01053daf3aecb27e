#!/bin/bash
# the executable's tests (layout decision for databases of several chunks among them), then soak seeds
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_soap4_cli.py -x -q -m gpu > gpurun_out/r4/t_cli.log 2>&1 || { tail -40 gpurun_out/r4/t_cli.log | cut -c1-600; exit 1; }
tail -2 gpurun_out/r4/t_cli.log
for sd in ${SOAK_SEEDS:-4611 4612}; do
  timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed $sd > gpurun_out/r4/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r4/soak_$sd.log | cut -c1-900; exit 1; }
  echo "seed $sd: $(tail -1 gpurun_out/r4/soak_$sd.log)"
done
