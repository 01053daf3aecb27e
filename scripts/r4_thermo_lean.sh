#!/bin/bash
# round 4: the certificate's new variants (tests), the lean seeding kernel (seed-stage tests + NT step A/B), the read-shape legs with / without the new variants
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py tests/test_gpu_parity.py -x -q > gpurun_out/r4/t_thermo.log 2>&1 || { tail -40 gpurun_out/r4/t_thermo.log; exit 1; }
tail -2 gpurun_out/r4/t_thermo.log
for lean in 1 0; do
  MP_SEED_LEAN=$lean timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/nt_lean$lean.json 2> gpurun_out/r4/nt_lean$lean.err || { tail -20 gpurun_out/r4/nt_lean$lean.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/nt_lean$lean.json") if l.startswith("{")][-1])
print("lean=$lean:", round(d["value"]), "pairs/s; single", round(d["single_context"]["pairs_per_s"]), "frac", round(d["roofline"]["frac"], 4), {k: round(v, 2) for k, v in d["kernel_ms_per_chunk_other_chunks"].items()})
PY
done
for more in 1 0; do
  MP_DP_THERMO_MORE=$more timeout -k 10 500 python bench.py --steps 1 --warmup 1 --nt-replica-gb 20 --nt-pairs 65536 --no-pmc --no-cpu-baseline --no-e2e --legs human_filter,C3_divergent,C4_2x250,C5_mixed_rescue,nt2_ini_same_reads > gpurun_out/r4/legs_more$more.json 2> gpurun_out/r4/legs_more$more.err || { tail -20 gpurun_out/r4/legs_more$more.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/legs_more$more.json") if l.startswith("{")][-1])
L = d["legs"]
print("thermo_more=$more: human", round(L["human_filter"]["pairs_per_s"]), {k: round(v, 2) for k, v in L["human_filter"]["kernel_ms_per_step"].items()})
for k in ("C3_divergent", "C4_2x250", "C5_mixed_rescue", "nt2_ini_same_reads"):
    print("   ", k, round(L[k].get("pairs_per_s", 0)), {a: round(b, 2) for a, b in L[k].get("kernel_ms_per_step", {}).items()}, L[k].get("error"))
PY
done
