# the round's closing check on the GPU box: the whole GPU test suite, the smoke test of __graft_entry__, a third soak seed
set -e
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_final.log 2>&1 || { tail -40 gpurun_out/r3/t_final.log; exit 1; }
tail -2 gpurun_out/r3/t_final.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 500 python3 scripts/soak_parity.py --rounds 40 --seed 124 > gpurun_out/r3/soak_124.log 2>&1 || { tail -5 gpurun_out/r3/soak_124.log | cut -c1-500; exit 1; }
tail -1 gpurun_out/r3/soak_124.log
