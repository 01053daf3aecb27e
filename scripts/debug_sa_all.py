"""Debug aid: every SA value / a sample of rank queries of a freshly built multi-sequence index, device vs oracle."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from megapath_amd import hipapi as H, index_builder, synth  # noqa: E402

tmp = tempfile.mkdtemp()
genome = synth.random_genome(600000, 501, n_seqs=40, repeat_frac=0.12)
arrs = index_builder.build_index_arrays(genome, sa_interval=16)
prefix = os.path.join(tmp, "ref.index")
index_builder.write_index_files(prefix, arrs)
lib = O.oracle()
oix = lib.mpo_index_open(prefix.encode())
n = lib.mpo_text_length(oix)
gix = H.Index(prefix, 0)
idx = np.arange(0, n + 1, dtype=np.uint64)
got = gix.sa(idx)
want = np.array([lib.mpo_sa_value(oix, int(i)) for i in idx], dtype=np.uint64)
bad = np.flatnonzero(got != want)
print("n", n, "inv_sa0", lib.mpo_inverse_sa0(oix), "SA mismatches", len(bad), bad[:10], got[bad[:10]], want[bad[:10]])
rng = np.random.default_rng(1)
qi = rng.integers(0, n + 2, size=200000).astype(np.uint64)
qc = rng.integers(0, 4, size=200000).astype(np.uint32)
gocc = gix.occ(qi, qc)
wocc = np.array([lib.mpo_occ(oix, int(i), int(c)) for i, c in zip(qi, qc)], dtype=np.uint64)
b2 = np.flatnonzero(gocc != wocc)
print("occ mismatches", len(b2), qi[b2[:10]], qc[b2[:10]], gocc[b2[:10]], wocc[b2[:10]])
keys = rng.integers(0, 4 ** 13, size=200000).astype(np.uint32)
gl, gr = gix.lkt(keys)
tab = lib.mpo_lkt_table(oix)
wl = np.array([tab[int(k) - 1] + 1 if k else 1 for k in keys], dtype=np.uint64)
wr = np.array([tab[int(k)] for k in keys], dtype=np.uint64)
print("lkt mismatches", int((gl != wl).sum() + (gr != wr).sum()))
