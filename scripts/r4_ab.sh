#!/bin/bash
# A/B on one box: the NT step with the library as built (A) and with megapath_amd/libmegapath_hip_ab.so (B, a build with one switch changed) in its place
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
run() {
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/ab_$1.json 2> gpurun_out/r4/ab_$1.err || { tail -20 gpurun_out/r4/ab_$1.err; exit 1; }
  python3 - $1 <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r4/ab_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]), round(d["single_context"]["ms_per_step"]), json.dumps(d["kernel_ms_per_chunk_other_chunks"]))
PY
}
run A
cp megapath_amd/libmegapath_hip.so /tmp/lib_a.so
cp megapath_amd/libmegapath_hip_ab.so megapath_amd/libmegapath_hip.so
run B
cp /tmp/lib_a.so megapath_amd/libmegapath_hip.so
run A2
