#!/bin/bash
# `python3 bench.py --gpus 2` by itself on the one-GPU box: both ranks on device 0 (MP_BENCH_SHARE_GPU=1), gloo instead of RCCL (RCCL refuses two ranks on one
# device; RCCL itself runs in tests/test_gpu_rccl.py), a small replica each (12 chunks of 400 Mbp) so that two of them fit side by side.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
env -u RANK -u LOCAL_RANK -u WORLD_SIZE MP_BENCH_SHARE_GPU=1 MP_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 --genome-mbp 400 --nt-replica-gb 5 --nt-pairs 65536 \
    > gpurun_out/r4/bench_gpus2.json 2> gpurun_out/r4/bench_gpus2.err || { tail -20 gpurun_out/r4/bench_gpus2.err; exit 1; }
python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_gpus2.json") if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value", round(d["value"]), "chunks", d["config"]["chunks"], d["hit_count_allreduce"])
PY
