#!/bin/bash
# round 4: whole-path soak against the oracle with the round's new switches drawn per round (-b with resident chunks through the device reduction or the
# host merge, the certificate's new variants on / off), the 2-rank rehearsal of bench.py --gpus 2 on one box, the profiles of the final build
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
for sd in ${SOAK_SEEDS:-4101 4102}; do
  timeout -k 10 400 python3 scripts/soak_parity.py --rounds 40 --seed $sd > gpurun_out/r4/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r4/soak_$sd.log | cut -c1-700; exit 1; }
  echo "seed $sd: $(tail -1 gpurun_out/r4/soak_$sd.log)"
done
bash scripts/r4_rehearse_2ranks.sh
bash scripts/r4_collect_profiles.sh r04_nt > gpurun_out/r4/collect.txt 2>&1 || { tail -20 gpurun_out/r4/collect.txt; exit 1; }
tail -3 gpurun_out/r4/collect.txt
