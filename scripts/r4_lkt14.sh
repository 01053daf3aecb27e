#!/bin/bash
# the packed 14-mer table (MP_LKT14): its tests, then the NT step with and without it
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_14mer or seed_stage_matches_oracle or fm_primitives" > gpurun_out/r4/t_lkt14.log 2>&1 || { tail -40 gpurun_out/r4/t_lkt14.log | cut -c1-600; exit 1; }
tail -2 gpurun_out/r4/t_lkt14.log
run() {
  name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-legs --no-pmc --no-cpu-baseline $ARGS > gpurun_out/r4/lk_$name.json 2> gpurun_out/r4/lk_$name.err || { tail -20 gpurun_out/r4/lk_$name.err | cut -c1-400; return 0; }
  python3 - $name <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r4/lk_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
c=d["config"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]), c["chunks"], round(c["index_bytes_in_hbm"]/1e9,1), "free", round(c["hbm_free_bytes_during_timed_steps"]/1e9,1), json.dumps(d["kernel_ms_per_chunk_other_chunks"]), flush=True)
PY
}
ARGS="--nt-pairs 196608 --nt-headroom-gb 38" run on192 MP_BENCH_NT_LKT14=1
ARGS="--nt-pairs 262144 --nt-headroom-gb 38" run on256 MP_BENCH_NT_LKT14=1
ARGS="--nt-pairs 196608" run off192 MP_BENCH_NT_LKT14=0
