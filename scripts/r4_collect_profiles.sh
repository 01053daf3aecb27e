#!/bin/bash
# Collects the rocprofv3 evidence of profiles/r04_* on the GPU box (run through gpurun from the repo root):  bash scripts/r4_collect_profiles.sh r04_nt
# 1. kernel trace + stats of the bench command in the headline's regime (one context, so that kernels do not overlap; 11 chunks instead of 89: the per-chunk
#    launches are the same, the run stays short); 2. SQ counters of the seed-stage and rescue kernels of one headline step (3 chunks) and of one human-filter step.
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) is collected by bench.py itself in child runs and lands in its JSON line (roofline.traffic_detail).
# rocprofv3 is started from /tmp with the program itself after `--`; --pmc passes are restricted to this repo's kernels with --kernel-include-regex.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r04_nt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --contexts 1 --nt-replica-gb 30 --no-cpu-baseline --no-pmc --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --kernel-include-regex "dp_fill_kernel|dp_score_kernel|dp_thermo_kernel|seed_kernel|merge_kernel|locate_kernel|expand_hits_kernel|pair_kernel|single_cand_kernel|floor_bound_kernel|floor_mate_kernel" --kernel-trace --output-format csv -d $OUT/valu -- python3 $ROOT/bench.py --pmc-child nt > $OUT/bench_valu.json 2> $OUT/valu.err
echo "valu nt done"
rocprofv3 --pmc $SQ --kernel-include-regex "dp_band_kernel|dp_fill_kernel|dp_score_kernel|dp_thermo_kernel|seed_kernel|merge_kernel|pair_kernel|dp_prefilter_kernel" --kernel-trace --output-format csv -d $OUT/valu_human -- python3 $ROOT/bench.py --pmc-child human > $OUT/bench_valu_human.json 2> $OUT/valu_human.err
echo "valu human done"
python3 $ROOT/scripts/summarize_profiles.py $OUT $TAG
ls -la $OUT/summary
# only the summaries travel back (gpurun merges at most 64 MiB of gpurun_out/)
rm -rf $OUT/trace $OUT/valu $OUT/valu_human
