#!/bin/bash
set -e
mkdir -p gpurun_out
for c in 3 4 6; do
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica --nt-replica-gb 60 --contexts $c > gpurun_out/nt_ctx_$c.json 2> gpurun_out/nt_ctx_$c.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/nt_ctx_$c.json") if l.startswith("{")][-1])
n = d["legs"]["nt_replica"]
print("contexts=$c nt", round(n["pairs_per_s"]), "single", round(n["single_context"]["pairs_per_s"]), "main value", round(d["value"]))
PY
done
