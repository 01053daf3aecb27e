#!/bin/bash
# one-off: the bench's parity sample at ten times its size (1 Mi pairs with soap4.ini, 512 Ki with soap4-nt2.ini) against the real soap4
set -e
mkdir -p gpurun_out/r3
timeout -k 10 1100 python bench.py --steps 2 --warmup 1 --no-pmc --no-legs --cpu-sample-pairs 1048576 > gpurun_out/r3/big_parity.json 2> gpurun_out/r3/big_parity.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r3/big_parity.json") if l.startswith("{")][-1])
print(d.get("parity_sample")); print(d.get("parity_sample_nt2_ini")); print(d["cpu_baseline"]["value"])
PY
