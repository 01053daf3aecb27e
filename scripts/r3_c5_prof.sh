#!/bin/bash
set -e
rm -rf gpurun_out/c5_prof; mkdir -p gpurun_out/c5_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5_prof -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pairs-per-step 65536 --no-cpu-baseline --no-pmc --no-e2e --legs C5_mixed_rescue --contexts 1 > $GRAFT_REPO_ROOT/gpurun_out/c5_prof/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/c5_prof/bench.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/c5_prof/c5_kernel_trace.csv
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/c5_prof/c5_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("dp_", "thermo", "rescue", "single_")) and float(r["TotalDurationNs"]) > 3e5:
        print("%-64s calls %6s avg_us %10.1f total_ms %9.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
d = json.loads([l for l in open("gpurun_out/c5_prof/bench.json") if l.startswith("{")][-1])
c = d["legs"]["C5_mixed_rescue"]
print(c["pairs_per_s"], c["pairs_per_step"], {k: round(v, 2) for k, v in c["kernel_ms_per_step"].items()})
PY
