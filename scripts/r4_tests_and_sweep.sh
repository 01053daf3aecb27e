#!/bin/bash
# round 4, first GPU call: the GPU suite after the clean-up (incl. the RCCL test), then the NT leg's batch-size sweep with the round-3 bench
set -e
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_all.log 2>&1 || { tail -40 gpurun_out/r4/t_all.log; exit 1; }
tail -3 gpurun_out/r4/t_all.log
for np in 65536 262144; do
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica --nt-pairs $np > gpurun_out/r4/nt_sweep_$np.json 2> gpurun_out/r4/nt_sweep_$np.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/nt_sweep_$np.json") if l.startswith("{")][-1])
n = d["legs"]["nt_replica"]
print($np, "nt pairs/s", n.get("pairs_per_s"), n.get("single_context"), n.get("kernel_ms_per_chunk_other_chunks"), n.get("error"))
PY
done
