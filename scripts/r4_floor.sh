#!/bin/bash
# round 4: the -top rule across resident chunks -- its tests, the chain goldens with it on (default now), a soak with the switch drawn, the headline with and without it
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_top_floor.py tests/test_gpu_soap4_cli.py -x -q -k "top_rule or floor or chunk or chain or bam" > gpurun_out/r4/t_floor.log 2>&1 || { tail -40 gpurun_out/r4/t_floor.log; exit 1; }
tail -2 gpurun_out/r4/t_floor.log
timeout -k 10 400 python3 scripts/soak_parity.py --rounds 30 --seed 4201 > gpurun_out/r4/soak_4201.log 2>&1 || { tail -5 gpurun_out/r4/soak_4201.log | cut -c1-700; exit 1; }
echo "soak 4201: $(tail -1 gpurun_out/r4/soak_4201.log)"
for fl in 1 0; do
  MP_TOP_FLOOR=$fl timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/nt_floor$fl.json 2> gpurun_out/r4/nt_floor$fl.err || { tail -20 gpurun_out/r4/nt_floor$fl.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/nt_floor$fl.json") if l.startswith("{")][-1])
print("floor=$fl:", round(d["value"]), "pairs/s; single", round(d["single_context"]["pairs_per_s"]), "chunks", d["config"]["chunks"], {k: round(v, 2) for k, v in d["kernel_ms_per_chunk_other_chunks"].items()})
print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["top_rule_across_chunks"].items() if k != "note"})
PY
done
