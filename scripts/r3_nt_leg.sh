#!/bin/bash
# the metric's configuration (legs.nt_replica) alone
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica > gpurun_out/nt_leg.json 2> gpurun_out/nt_leg.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/nt_leg.json") if l.startswith("{")][-1])
n = d["legs"]["nt_replica"]
print("main value", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
print(n.get("rescue_certificate_per_pair_per_chunk")); print("nt", n["pairs_per_s"], n["single_context"], n["kernel_ms_per_chunk_other_chunks"], n["roofline"]["frac"])
print(d.get("dp", {}).get("rescue_certificate_per_step"))
PY
