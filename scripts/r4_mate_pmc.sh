#!/bin/bash
# SQ counters of the -top rule's kernels in one headline step (3 chunks)
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_mate
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-include-regex "floor_mate_kernel|floor_bound_kernel" --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --pmc-child nt > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-include-regex "floor_mate_kernel|floor_bound_kernel" --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --pmc-child nt > $OUT/b.json 2> $OUT/b.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for sub in "a","b":
    f=glob.glob(out+"/"+sub+"/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
rm -rf $OUT/a $OUT/b
