#!/bin/bash
# round 4: whole GPU suite on the current build, then the headline without legs (two-walk locate kernel, adaptive mate-rescue certificate)
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_all2.log 2>&1 || { tail -40 gpurun_out/r4/t_all2.log; exit 1; }
tail -2 gpurun_out/r4/t_all2.log
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-legs --no-pmc --no-cpu-baseline > gpurun_out/r4/nt_loc2.json 2> gpurun_out/r4/nt_loc2.err || { tail -20 gpurun_out/r4/nt_loc2.err; exit 1; }
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/nt_loc2.json") if l.startswith("{")][-1])
print("two-walk locate:", round(d["value"]), "pairs/s; single", round(d["single_context"]["pairs_per_s"]), "frac", round(d["roofline"]["frac"], 4), {k: round(v, 2) for k, v in d["kernel_ms_per_chunk_other_chunks"].items()})
PY
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
