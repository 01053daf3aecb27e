#!/bin/bash
# the executable with gzip'ed input: threads per stream
set -e
mkdir -p gpurun_out
export MP_BENCH_GZ_VARIANTS="t3:MP_INFLATE_THREADS=3;t4:MP_INFLATE_THREADS=4;t8:MP_INFLATE_THREADS=8;t6r4:MP_INFLATE_THREADS=6,MP_READER_THREADS=4"
timeout -k 10 1000 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --legs e2e_cli --e2e-batches 8 > gpurun_out/gz_e2e.json 2> gpurun_out/gz_e2e.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/gz_e2e.json") if l.startswith("{")][-1])
e = d["legs"]["e2e_cli"]
for k, v in e.items():
    if isinstance(v, dict) and "pairs_per_s" in v:
        print(k, round(v["pairs_per_s"] / 1e6, 2), (v.get("stage_seconds") or "")[:110])
PY
