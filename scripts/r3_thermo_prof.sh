#!/bin/bash
set -e
mkdir -p gpurun_out/thermo_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/thermo_prof -o nt2 -- python3 $GRAFT_REPO_ROOT/bench.py --ini soap4-nt2.ini --L 151 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --no-legs --contexts 1 > $GRAFT_REPO_ROOT/gpurun_out/thermo_prof/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/thermo_prof/bench.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/thermo_prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s avg_us %10.1f  pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
