#!/bin/bash
# certificate kernel: parity tests, then the nt2 configuration (one context: kernel split)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py -x -q > gpurun_out/thermo_tests.log 2>&1 || { tail -30 gpurun_out/thermo_tests.log; exit 1; }
tail -2 gpurun_out/thermo_tests.log
timeout -k 10 600 python bench.py --ini soap4-nt2.ini --L 151 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs C3_divergent,C4_2x250,C5_mixed_rescue --contexts 1 > gpurun_out/thermo_q.json 2> gpurun_out/thermo_q.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/thermo_q.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"])
print(" split", {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
c3 = d.get("legs", {}).get("C3_divergent", {})
print(" C3", c3.get("pairs_per_s"), {k: round(v, 2) for k, v in c3.get("kernel_ms_per_step", {}).items()})
PY
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/thermo_q.json") if l.startswith("{")][-1])
for k in ("C4_2x250", "C5_mixed_rescue"):
    c = d.get("legs", {}).get(k, {})
    print(k, c.get("pairs_per_s"), {a: round(b, 2) for a, b in c.get("kernel_ms_per_step", {}).items()}, c.get("error"))
PY
