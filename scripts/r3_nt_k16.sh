#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "16mer or seed_stage" > gpurun_out/k16_tests.log 2>&1 || { tail -30 gpurun_out/k16_tests.log; exit 1; }
tail -2 gpurun_out/k16_tests.log
for k in 0 16; do
  MP_BENCH_NT_KMER=$k timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --legs nt_replica --nt-replica-gb 60 > gpurun_out/nt_k_$k.json 2> gpurun_out/nt_k_$k.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/nt_k_$k.json") if l.startswith("{")][-1])
n = d["legs"]["nt_replica"]
print("kmer=$k chunks", n["chunks"], "nt", round(n["pairs_per_s"]), "single", round(n["single_context"]["pairs_per_s"]), "chunk-pairs/s", round(n["chunk_pair_alignments_per_s"]), {k: round(v, 2) for k, v in n["kernel_ms_per_chunk_other_chunks"].items() if k in ("seed", "locate", "rescue_fill")}, n["roofline"]["frac"])
PY
done
