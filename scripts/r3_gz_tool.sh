#!/bin/bash
# raw rates of the gzip readers on the box's cores (no GPU work)
set -e
mkdir -p gpurun_out /dev/shm/pgz
g++ -O2 -std=c++17 -pthread -o gpurun_out/pargz_check tests/tools/pargz_check.cpp -lz
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
n = 1_000_000
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
q = np.frombuffer(b"FFFFF:,#", dtype=np.uint8)[rng.integers(0, 8, (n, 150))]
with open("/dev/shm/pgz/r.fq", "wb") as f:
    for a in range(0, n, 100000):
        out = []
        for i in range(a, min(n, a + 100000)):
            out.append(b"@read%d/1\n" % i + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
        f.write(b"".join(out))
PY
gzip -1 -k /dev/shm/pgz/r.fq
ls -la /dev/shm/pgz
for t in 1 2 3 4 6 8 12; do gpurun_out/pargz_check /dev/shm/pgz/r.fq.gz $t 4194304 2 | grep -E "sequential|parallel|thread-seconds" | tail -3; done > gpurun_out/gz_tool.log 2>&1
rm -rf /dev/shm/pgz gpurun_out/pargz_check
cat gpurun_out/gz_tool.log
