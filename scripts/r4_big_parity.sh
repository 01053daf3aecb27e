#!/bin/bash
# one-off: the headline's parity sample at ten times its default size (200 000 pairs: the real soap4 chained over chunk 0 + one foreign 2.95 Gbp chunk with the
# NT stage's command lines, against bin/soap4-mi355x with both chunks resident), on a small replica so that the run stays short
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 1100 python3 bench.py --steps 1 --warmup 1 --nt-replica-gb 12 --nt-cpu-pairs 200000 --no-pmc --no-legs > gpurun_out/r4/big_parity.json 2> gpurun_out/r4/big_parity.err; echo "rc $?"
python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/big_parity.json") if l.startswith("{")][-1])
p = d["parity_sample"]
print({k: p.get(k) for k in ("pairs", "records", "records_equal", "records_differing", "reference_records_with_a_hit", "reference_records_listing_two_or_more_references", "first_diff")})
print(p.get("how"))
print(d["cpu_baseline"])
PY
