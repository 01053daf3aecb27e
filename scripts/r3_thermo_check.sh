#!/bin/bash
# thermo certificate: parity tests, then the nt2 configuration with the certificate on and off (one context: kernel split), then C3 + a small NT replica
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py -x -q > gpurun_out/thermo_tests.log 2>&1 || { tail -30 gpurun_out/thermo_tests.log; exit 1; }
tail -3 gpurun_out/thermo_tests.log
for on in 1 0; do
  MP_DP_THERMO=$on timeout -k 10 600 python bench.py --ini soap4-nt2.ini --L 151 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs C3_divergent > gpurun_out/thermo_nt2_$on.json 2> gpurun_out/thermo_nt2_$on.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/thermo_nt2_$on.json") if l.startswith("{")][-1])
print("thermo=$on value", d["value"], "ms", d["ms_per_step"])
print(" split", d.get("kernel_ms_per_step") or d.get("detail", {}).get("kernel_ms_per_step"))
print(" C3", {k: v for k, v in d.get("legs", {}).get("C3_divergent", {}).items() if k in ("value", "pairs_per_s", "ms_per_step", "kernel_ms_per_step")})
PY
done
