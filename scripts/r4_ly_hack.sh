#!/bin/bash
# measurement only (unsound switch): what a bound on the rescued mate's score below its length would be worth in the -top rule
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
for ly in 0 136 125 112; do
  MP_FLOOR_HACK_LY=$ly timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/ly_$ly.json 2> gpurun_out/r4/ly_$ly.err || { tail -20 gpurun_out/r4/ly_$ly.err; exit 1; }
  python3 - $ly <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r4/ly_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]), d["top_rule_across_chunks"]["share_of_failed_pairs_left_out"], d["kernel_ms_per_chunk_other_chunks"]["rescue_fill"], d["kernel_ms_per_chunk_other_chunks"]["rescue_tasks"])
PY
done
