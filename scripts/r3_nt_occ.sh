#!/bin/bash
set -e
mkdir -p gpurun_out
for o in 2 1; do
  export MP_THERMO_OCC=$o
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica --nt-replica-gb 60 > gpurun_out/nt_occ_$o.json 2> gpurun_out/nt_occ_$o.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/nt_occ_$o.json") if l.startswith("{")][-1])
n = d["legs"]["nt_replica"]
print("occ=$o nt", n["pairs_per_s"], n["single_context"]["pairs_per_s"], {k: round(v, 2) for k, v in n["kernel_ms_per_chunk_other_chunks"].items()})
PY
done
