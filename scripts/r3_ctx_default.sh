#!/bin/bash
set -e
mkdir -p gpurun_out
for c in 2 3 4; do
  timeout -k 10 600 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --no-legs --contexts $c > gpurun_out/ctxd_$c.json 2> gpurun_out/ctxd_$c.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/ctxd_$c.json") if l.startswith("{")][-1])
print("contexts=$c value", round(d["value"]), "ms", round(d["ms_per_step"], 2), "single", round(d["single_context"]["pairs_per_s"]))
PY
done
