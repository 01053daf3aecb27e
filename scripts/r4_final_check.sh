#!/bin/bash
# round 4, final build: the whole GPU suite, two more soak seeds (every switch of the round drawn per round), then the driver's own bench command
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_all3.log 2>&1 || { tail -40 gpurun_out/r4/t_all3.log; exit 1; }
tail -2 gpurun_out/r4/t_all3.log
SOAK_SEEDS="4301 4302" 
for sd in 4301 4302; do
  timeout -k 10 400 python3 scripts/soak_parity.py --rounds 40 --seed $sd > gpurun_out/r4/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r4/soak_$sd.log | cut -c1-700; exit 1; }
  echo "seed $sd: $(tail -1 gpurun_out/r4/soak_$sd.log)"
done
