#!/bin/bash
# kernel trace of the headline's regime only (pass 1 of r4_collect_profiles.sh) and its summaries
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r04_nt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --contexts 1 --nt-replica-gb 30 --no-cpu-baseline --no-pmc --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
python3 $ROOT/scripts/summarize_profiles.py $OUT $TAG
rm -rf $OUT/trace
