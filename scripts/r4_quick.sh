#!/bin/bash
# a change in the rescue path: the tests around it, one soak seed, the NT step
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_soap4_cli.py -x -q -m gpu > gpurun_out/r4/t_quick.log 2>&1 || { tail -40 gpurun_out/r4/t_quick.log | cut -c1-500; exit 1; }
tail -2 gpurun_out/r4/t_quick.log
timeout -k 10 400 python3 scripts/soak_parity.py --rounds 30 --seed ${SOAK_SEED:-4701} > gpurun_out/r4/soak_q.log 2>&1 || { tail -5 gpurun_out/r4/soak_q.log | cut -c1-900; exit 1; }
echo "soak: $(tail -1 gpurun_out/r4/soak_q.log)"
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/quick.json 2> gpurun_out/r4/quick.err || { tail -20 gpurun_out/r4/quick.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4/quick.json") if l.startswith("{")][-1])
print(round(d["value"]), d.get("parity_ok"), d["top_rule_across_chunks"]["share_of_failed_pairs_left_out"], json.dumps(d["kernel_ms_per_chunk_other_chunks"]))
PY
