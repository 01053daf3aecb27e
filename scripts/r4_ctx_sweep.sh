#!/bin/bash
# contexts / batch size after the -top rule's second half (less rescue work per chunk: does the balance move?)
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
for cfg in ${SWEEP:-"4 262144" "5 262144" "6 262144" "4 327680"}; do
  set -- $cfg
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --contexts $1 --nt-pairs $2 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/ctx_$1_$2.json 2> gpurun_out/r4/ctx_$1_$2.err || { tail -5 gpurun_out/r4/ctx_$1_$2.err; continue; }
  python3 - $1 $2 <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r4/ctx_%s_%s.json" % (sys.argv[1], sys.argv[2])) if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], round(d["value"]), round(d["ms_per_step"]), d.get("parity_ok"), "free GB", d["config"]["hbm_free_bytes_during_timed_steps"]/1e9, "single", round(d["single_context"]["ms_per_step"]))
PY
done
