# three more soak seeds on the final build (ini, DP switches, merge rows, table combinations, chunking all drawn per round)
set -e
mkdir -p gpurun_out/r3
for sd in 9601 9602 9603; do
  timeout -k 10 330 python3 scripts/soak_parity.py --rounds 40 --seed $sd > gpurun_out/r3/soak_$sd.log 2>&1 || { tail -5 gpurun_out/r3/soak_$sd.log | cut -c1-600; exit 1; }
  echo "seed $sd: $(tail -1 gpurun_out/r3/soak_$sd.log)"
done
