set -e
OUT=$PWD/gpurun_out/r3/tl; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-legs --contexts 3 > $OUT/b3.json 2> $OUT/b3.err
python3 $ROOT/scripts/timeline.py $(find $OUT/trace3 -name "*kernel_trace.csv" | head -1) 12 | tee $OUT/timeline3.txt
