mkdir -p gpurun_out/r3
for sd in ${SOAK_SEEDS:-411 977}; do
  timeout -k 10 520 python3 scripts/soak_parity.py --rounds 45 --seed $sd > gpurun_out/r3/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r3/soak_$sd.log | cut -c1-600; exit 1; }
  echo "seed $sd: $(grep -c '^round' gpurun_out/r3/soak_$sd.log) rounds, $(tail -1 gpurun_out/r3/soak_$sd.log | cut -c1-100)"
done
