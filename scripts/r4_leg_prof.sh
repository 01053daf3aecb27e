#!/bin/bash
# kernel-level breakdown of one read-shape leg with the certificate's variants on (1) and at round 3's coverage (2)
set -e
LEG=${1:-C5_mixed_rescue}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4/legprof_$LEG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lv in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l$lv -o k -- python3 $GRAFT_REPO_ROOT/scripts/r4_leg_prof.py $LEG $lv > $OUT/l$lv.out 2> $OUT/l$lv.err || { tail -20 $OUT/l$lv.err; exit 1; }
  tail -1 $OUT/l$lv.out
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/l$lv/**/k_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    if any(k in r["Name"] for k in ("dp_", "thermo", "rescue", "single_", "seed_", "merge_", "pair_", "locate", "expand", "Scan", "scan", "failed")) and float(r["TotalDurationNs"]) > 1.5e6:
        print("  %-110s calls %5s avg_us %9.1f total_ms %8.1f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $OUT/l$lv
done
