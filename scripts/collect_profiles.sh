#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r01
# 1. kernel trace + stats of the default bench command; 2./3. FETCH_SIZE and WRITE_SIZE in their own passes; 4. SQ counters of the DP kernels; restricted
# to this repo's kernels (a --pmc pass over the torch kernels of the index build crashes rocprofv3 on this image).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
RX="gather64_kernel|seed_kernel|seed_overflow_kernel|locate_kernel|expand_hits_kernel|dp_fill_kernel|dp_band_kernel|dp_score_kernel"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
echo "write done"
# 4. VALU counters of the DP kernels (SQ block only, with the kernel trace for their durations)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "dp_band_kernel|dp_fill_kernel|dp_score_kernel" --kernel-trace --output-format csv -d $OUT/valu -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_valu.json 2> $OUT/valu.err
echo "valu done"
python3 scripts/summarize_profiles.py $OUT $TAG
