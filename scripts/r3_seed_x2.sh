#!/bin/bash
set -e
mkdir -p gpurun_out
MP_SEED_ITEMS=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "seed or occ or prefilter or merge" > gpurun_out/x2_tests.log 2>&1 || { tail -30 gpurun_out/x2_tests.log; exit 1; }
tail -2 gpurun_out/x2_tests.log
for it in 1 2; do
  MP_SEED_ITEMS=$it timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --no-legs --contexts 1 > gpurun_out/x2_$it.json 2> gpurun_out/x2_$it.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/x2_$it.json") if l.startswith("{")][-1])
print("items=$it value", round(d["value"]), "ms", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items() if k in ("seed", "merge", "fill")}, "frac", round(d["roofline"]["frac"], 4))
PY
done
