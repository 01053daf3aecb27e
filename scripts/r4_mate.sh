#!/bin/bash
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_top_floor.py -x -q > gpurun_out/r4/t_mate.log 2>&1 || { tail -40 gpurun_out/r4/t_mate.log | cut -c1-600; exit 1; }
tail -2 gpurun_out/r4/t_mate.log
for v in 1 0; do
MP_TOP_FLOOR_MATE=$v timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/mate_$v.json 2> gpurun_out/r4/mate_$v.err || { tail -20 gpurun_out/r4/mate_$v.err; exit 1; }
python3 - $v <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r4/mate_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]), d.get("parity_ok"), d["top_rule_across_chunks"]["share_of_failed_pairs_left_out"], json.dumps(d["kernel_ms_per_chunk_other_chunks"]))
PY
done
