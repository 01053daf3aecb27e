mkdir -p gpurun_out/r3
timeout -k 10 1000 python3 scripts/soak_parity.py --rounds 70 --seed 31 > gpurun_out/r3/soak_31.log 2>&1
rc=$?
grep -c "^round" gpurun_out/r3/soak_31.log
tail -4 gpurun_out/r3/soak_31.log | cut -c1-600
exit $rc
