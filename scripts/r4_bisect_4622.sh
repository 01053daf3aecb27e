#!/bin/bash
# soak seed 4622, round 15 (BAM unpair records differ): the round as drawn, then with single switches changed
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
for v in "" "MP_LKT14=0" "MP_TOP_FLOOR=0" "MP_STASH_PLAIN=0" "MP_LKT14=0,MP_TOP_FLOOR=0"; do
  echo "== env override: '$v'"
  timeout -k 10 200 python3 scripts/soak_parity.py --rounds 16 --seed 4622 --only 15 --env "$v" 2>&1 | grep -v "^round" | cut -c1-700 | tail -8
done
