import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from megapath_amd import hipapi as H
from megapath_amd.index_builder import build_index_arrays
v = np.load(os.path.join(ROOT, "tests/golden/dp_vectors.npz"))
ro = np.concatenate([[0], np.cumsum(v["ref_len"])]); qo = np.concatenate([[0], np.cumsum(v["read_len"])])
arrs = build_index_arrays([("cat", v["ref_cat"].astype(np.uint8))], sa_interval=16)
ix = H.Index.from_arrays(0, **arrs); ctx = H.Context(ix)
def run(sel, clip, label):
    seqs = []
    for t in sel:
        s = bytes(b"ACGT"[c] for c in v["read_cat"][qo[t]:qo[t + 1]]); seqs += [s, s]
    L = max(len(s) for s in seqs) + 1
    packed, lens, W = H.pack_reads(seqs, L); ctx.set_reads(packed, lens)
    tasks = [H.DpTask(2 * k, 1, int(ro[t]), int(v["ref_len"][t]), 0) for k, t in enumerate(sel)]
    res, ops, stride, _ = ctx.sgdp_batch(tasks, clip[0], clip[1], want_ops=True)
    for k, t in enumerate(sel):
        cig = ""
        if res[k].score > 0:
            buf = C.create_string_buffer(4 * stride + 16); o = np.ascontiguousarray(ops[k * stride:k * stride + res[k].n_ops])
            O.oracle().mpo_cigar_from_ops(O.as_u8p(o), res[k].n_ops, buf, len(buf)); cig = buf.value.decode()
        got = (res[k].score, res[k].hit_loc, res[k].max_score_count, cig)
        want = (int(v["score"][t]), int(v["hit_loc"][t]), int(v["count"][t]), str(v["cigar"][t]))
        print(label, k, t, "OK" if got == want else "DIFF", got if got != want else "", want if got != want else "")
run([0], (130, 130), "single")
run([0, 0], (130, 130), "dup")
run([0, 1, 2, 3], (130, 130), "four")
run([320, 321], (0, 0), "clip0")
run([320], (0, 0), "clip0single")
