#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r02
# 1. kernel trace + stats of the bench command (one context, so that kernels do not overlap); 2. SQ counters of the DP and the
# seed-stage kernels (SQ block only, with the kernel trace for their durations).  HBM traffic (FETCH_SIZE / WRITE_SIZE, separate
# passes) is collected by bench.py itself in two child runs and lands in its JSON line (roofline.traffic_detail).
# rocprofv3 is started from /tmp with the program itself after `--`; --pmc passes are restricted to this repo's kernels with
# --kernel-include-regex: a --pmc pass over the whole process dies with SIGSEGV inside rocprofv3 when it reaches a torch
# device-to-device copy (at::native::copy_device_to_device, the slice assignment of simulate.genome).
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-legs --contexts 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "dp_band_kernel|dp_fill_kernel|dp_score_kernel|dp_thermo_kernel|seed_kernel|merge_kernel|locate_kernel|expand_hits_kernel|pair_kernel" --kernel-trace --output-format csv -d $OUT/valu -- python3 $ROOT/bench.py --pmc-child --steps 1 --warmup 0 > $OUT/bench_valu.json 2> $OUT/valu.err
echo "valu done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "dp_score_kernel|dp_thermo_kernel|dp_fill_kernel" --kernel-trace --output-format csv -d $OUT/valu_nt2 -- python3 $ROOT/bench.py --pmc-child --steps 1 --warmup 0 --ini soap4-nt2.ini --L 151 > $OUT/bench_valu_nt2.json 2> $OUT/valu_nt2.err
echo "valu nt2 done"
python3 $ROOT/scripts/summarize_profiles.py $OUT $TAG
