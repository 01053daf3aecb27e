"""Certificates of one instance set under the library as built: writes the cert arrays of every variant to an .npz (compared across builds by the caller)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_thermo as T
from megapath_amd import hipapi as H, index_builder

out = sys.argv[1]
res = {}
for name, lo, hi, margin, thermo_kind, rescue in (("narrow", 101, 160, 30, 1, False), ("short", 45, 100, 25, 0, False), ("wide", 161, 320, 30, 2, False),
                                                    ("rescue", 75, 160, 0, 3, True), ("rescue_wide", 100, 320, 0, 4, True)):
    rng = np.random.default_rng(99)
    text = rng.integers(0, 4, 1 << 21).astype(np.uint8)
    n = 40000
    if rescue:
        reads, tasks, kind = T._make_rescue_instances(rng, text, n, (0, 0, 0, 1, 2, 3, 5, 6), lo, hi)
    else:
        reads, tasks, kind = T._make_instances(rng, text, n, (0, 0, 0, 1, 2, 3, 4, 5, 6, 7), lo, hi, margin)
    arrs = index_builder.build_index_arrays([("t", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    seqs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    packed, lens, W = H.pack_reads(seqs, 161 if hi <= 160 else 321)
    ctx.set_reads(packed, lens)
    for rep in range(3):
        cert, st = ctx.test_dp_thermo(np.ascontiguousarray(tasks), thermo_kind)
        res["%s_%d" % (name, rep)] = cert.copy()
        print(name, rep, st, flush=True)
    res[name + "_kind"] = kind; res[name + "_len"] = lens[tasks["read_id"]]; res[name + "_m"] = tasks["win_len"]
    ctx.close(); ix.close()
np.savez(out, **res)
