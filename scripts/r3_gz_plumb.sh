#!/bin/bash
# the reader alone (MP_PLUMBING_TEST: nothing is aligned) on gzip'ed input: where does a batch's time go?
set -e
mkdir -p gpurun_out /dev/shm/pgz
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
n = 3_000_000
for k in (1, 2):
    with open("/dev/shm/pgz/r_%d.fq" % k, "wb") as f:
        for a in range(0, n, 250000):
            m = min(n, a + 250000) - a
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (m, 150))]
            q = np.frombuffer(b"FFFFF:,#", dtype=np.uint8)[rng.integers(0, 8, (m, 150))]
            f.write(b"".join(b"@read%d/%d\n" % (a + i, k) + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(m)))
PY
gzip -1 -k /dev/shm/pgz/r_1.fq & gzip -1 -k /dev/shm/pgz/r_2.fq & wait
ls -la /dev/shm/pgz
run() { tag=$1; shift; /usr/bin/env "$@" MP_PLUMBING_TEST=1 MP_TIMING=1 bin/soap4-mi355x pair x /dev/shm/pgz/r_1.fq$SUF /dev/shm/pgz/r_2.fq$SUF -o /dev/shm/pgz/o -C tests/soap4-nt2.ini -L 151 -T 16 -u 750 -F -nc > /dev/null 2> gpurun_out/plumb_$tag.err; echo "== $tag"; grep -E "reader:|alignment loop" gpurun_out/plumb_$tag.err | tail -4; }
SUF="" run plain X=1
SUF=".gz" run gz_seq MP_PARGZ_MIN=99999999999
SUF=".gz" run gz_par3 MP_INFLATE_THREADS=3
SUF=".gz" run gz_par6 MP_INFLATE_THREADS=6
rm -rf /dev/shm/pgz
