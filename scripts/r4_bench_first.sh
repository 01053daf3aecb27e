#!/bin/bash
# round 4: the -b path through the device reduction (CLI goldens), then the restructured bench (NT headline) without legs, then the batch-size / context sweep
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_soap4_cli.py -x -q -k "bam or chunk or chain" > gpurun_out/r4/t_cli.log 2>&1 || { tail -40 gpurun_out/r4/t_cli.log; exit 1; }
tail -2 gpurun_out/r4/t_cli.log
for cfg in "262144 3" "524288 3" "786432 3" "524288 2" "1048576 4"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline --nt-pairs $1 --contexts $2 > gpurun_out/r4/nt_$1_$2.json 2> gpurun_out/r4/nt_$1_$2.err || { tail -20 gpurun_out/r4/nt_$1_$2.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/nt_$1_$2.json") if l.startswith("{")][-1])
print("$1 x $2:", round(d["value"]), "pairs/s", round(d["ms_per_step"]), "ms/step; single", round(d["single_context"]["pairs_per_s"]), "frac", round(d["roofline"]["frac"], 4),
      {k: round(v / ($1 / 65536), 2) for k, v in d["kernel_ms_per_chunk_other_chunks"].items()})
PY
done
