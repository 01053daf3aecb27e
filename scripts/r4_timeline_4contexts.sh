#!/bin/bash
# kernel timeline of the NT step with its four contexts (12 chunks: the per-chunk launches are the same as with 89): how much of the window some kernel runs, how many at once
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4/tl; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace4 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --nt-replica-gb 30 --no-cpu-baseline --no-pmc --no-legs --contexts 4 > $OUT/b4.json 2> $OUT/b4.err
python3 $ROOT/scripts/timeline.py $(find $OUT/trace4 -name "*kernel_trace.csv" | head -1) 160 | tee $OUT/timeline4.txt
rm -rf $OUT/trace4
