#!/bin/bash
# the whole GPU suite, then soak seeds (every switch of the round drawn per round, chunked databases with diverged copies)
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_all4.log 2>&1 || { tail -40 gpurun_out/r4/t_all4.log | cut -c1-500; exit 1; }
tail -2 gpurun_out/r4/t_all4.log
for sd in ${SOAK_SEEDS:-4501 4502}; do
  timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed $sd > gpurun_out/r4/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r4/soak_$sd.log | cut -c1-900; exit 1; }
  echo "seed $sd: $(tail -1 gpurun_out/r4/soak_$sd.log)"
done
