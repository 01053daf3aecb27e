"""Debug aid: which pairs does the product align in the deep-DP round that the oracle does not (or vice versa)?"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from megapath_amd import hipapi as H, index_builder, pipeline, synth  # noqa: E402

tmp = tempfile.mkdtemp()
genome = synth.random_genome(600000, 501, n_seqs=40, repeat_frac=0.12)
arrs = index_builder.build_index_arrays(genome, sa_interval=16)
prefix = os.path.join(tmp, "ref.index")
index_builder.write_index_files(prefix, arrs)
names, r1, r2 = synth.simulate_pairs(genome, 3000, 150, 502, sub=0.02, ins=0.002, dele=0.002, random_frac=0.03, broken_mate_frac=0.05,
                                     noisy_mate_frac=0.4, noisy_sub=0.2, n_frac=0.001)
f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
synth.write_fastq_pair(f1, f2, names, r1, r2)
lib = O.oracle()
oix = lib.mpo_index_open(prefix.encode())
p = O.default_params(L=151)
st = O.AlignStats()
rec = os.path.join(tmp, "o.rec")
assert lib.mpo_align_fastq(oix, C.byref(p), f1.encode(), f2.encode(), os.path.join(tmp, "o.fq").encode(), rec.encode(), C.byref(st)) == 0
odeep = set()
for line in open(rec):
    f = line.split("\t")
    if f[2] == "deep":
        odeep.add(int(f[0][1:]) - 1)
gix = H.Index(prefix, 0)
ctx = H.Context(gix)
recs = list(zip(pipeline.read_fastq(f1), pipeline.read_fastq(f2)))
flat = [x for pr in recs for x in pr]
packed, lens, _W = H.pack_reads([r[2] for r in flat], 151)
ctx.set_reads(packed, lens)
ap = H.AlignParams(max(1, 150), 750, 130, 130, H.MmpParams.nt2(), 1, 0)
hits, nh, rst = ctx.deep_dp_round(ap)
a = np.frombuffer(hits, dtype=np.dtype([("pair", "<u4"), ("lm", "<u4"), ("a1", "<u8"), ("a2", "<u8"), ("s1", "<i4"), ("s2", "<i4"),
                                        ("st1", "<u4"), ("st2", "<u4"), ("d1", "<u2"), ("d2", "<u2"), ("c1", "<u2"), ("c2", "<u2")]), count=nh)
gdeep = set(a["pair"].tolist())
print("oracle deep", len(odeep), "product deep", len(gdeep))
for q in sorted(gdeep ^ odeep):
    print("pair", q, "in product" if q in gdeep else "in oracle")
    for h in a[a["pair"] == q]:
        print("   hit", h)
    # oracle view of that pair: seeds
    for r in (2 * q, 2 * q + 1):
        codes = H._CODE[np.frombuffer(flat[r][2], dtype=np.uint8)].astype(np.uint8)
        cap = 4096
        pos = (O.SeedPos * cap)(); neg = (O.SeedPos * cap)(); npos = C.c_int(); nneg = C.c_int(); ctr = O.SeedCounters()
        lib.mpo_seed_read(oix, O.as_u8p(codes), len(codes), r & ~1, C.byref(p.mmp), pos, C.byref(npos), neg, C.byref(nneg), cap, C.byref(ctr))
        print("   oracle seeds read", r, "len", len(codes), "+", [(pos[i].pos, pos[i].paired_seed_length) for i in range(npos.value)],
              "-", [(neg[i].pos, neg[i].paired_seed_length) for i in range(nneg.value)])
    ctx2 = ctx
    seeds, off = ctx2.get_seeds()
    for r in (2 * q, 2 * q + 1):
        print("   product seeds read", r, [(int(seeds[i].pos), hex(int(seeds[i].strand_read_id)), int(seeds[i].paired_seed_length)) for i in range(int(off[r]), int(off[r + 1]))])
