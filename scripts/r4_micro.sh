#!/bin/bash
# round 4: two small changes on the NT step (certificate sweep ends at the last lane that owns read columns; single-end candidates sorted in LDS):
# the whole GPU suite, one soak seed, then the NT step without legs
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_micro.log 2>&1 || { tail -40 gpurun_out/r4/t_micro.log; exit 1; }
tail -2 gpurun_out/r4/t_micro.log
timeout -k 10 400 python3 scripts/soak_parity.py --rounds 30 --seed 4401 > gpurun_out/r4/soak_4401.log 2>&1 || { tail -5 gpurun_out/r4/soak_4401.log | cut -c1-700; exit 1; }
echo "soak: $(tail -1 gpurun_out/r4/soak_4401.log)"
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/bench_micro.json 2> gpurun_out/r4/bench_micro.err || { tail -20 gpurun_out/r4/bench_micro.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4/bench_micro.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("parity_ok"), json.dumps(d["roofline"])[:400])
PY
