#!/bin/bash
# the driver's own command line, timed; the JSON line is kept for profiles/
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
/usr/bin/env time -v true 2>/dev/null || true
START=$(date +%s)
timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_full.json 2> gpurun_out/r4/bench_full.err; RC=$?
END=$(date +%s)
echo "rc $RC, $((END - START)) s"
grep "^\[bench\]" gpurun_out/r4/bench_full.err | cut -c1-300
python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_full.json") if l.startswith("{")][-1])
print("value", round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"]), "parity_ok", d.get("parity_ok"))
print("roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k in ("achieved", "frac", "traffic", "launch_ms", "traffic_over_requested", "frac_of_gather_ceiling", "traffic_gbs")})
print("cpu", d.get("cpu_baseline", {}) and {k: d["cpu_baseline"][k] for k in ("value", "cores", "pairs_per_s_one_chunk")})
print("parity", {k: d["parity_sample"].get(k) for k in ("records", "records_equal", "records_differing", "reference_records_with_a_hit", "reference_records_listing_two_or_more_references", "first_diff")} if d.get("parity_sample") else None)
L = d.get("legs", {})
for k, v in L.items():
    if k == "e2e_cli":
        print("e2e", {a: (round(b["pairs_per_s"]) if isinstance(b, dict) and b.get("pairs_per_s") else None) for a, b in v.items() if isinstance(b, dict) and "pairs_per_s" in b}, v.get("host_cpus"))
        rc = v.get("resident_chunks", {})
        print("   resident", {a: (b.get("pairs_per_s") and round(b["pairs_per_s"]), b.get("stage_seconds", "")[-60:]) for a, b in rc.items() if isinstance(b, dict)})
    else:
        print(k, round(v.get("pairs_per_s", 0)), v.get("error"), (v.get("parity_sample") or {}).get("records_equal") if k == "human_filter" else "")
PY
