#!/bin/bash
set -e
rm -rf gpurun_out/nt_prof; mkdir -p gpurun_out/nt_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nt_prof -o nt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica --nt-replica-gb 30 --contexts 1 > $GRAFT_REPO_ROOT/gpurun_out/nt_prof/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/nt_prof/bench.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/nt_prof/nt_kernel_trace.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/nt_prof/nt_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("dp_", "thermo", "seed_kernel", "single_")):
        print("%-64s calls %6s avg_us %10.1f total_ms %9.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
