"""One-off stress of the exactness claims behind the banded DP kernels (run on the GPU box):
    python scripts/band_stress.py [--rounds 12]
Random error profiles, read lengths and genomes with tandem repeats (several copies of a unit inside one DP window: the
first-maximum rule then has to pick among equal alignments); the deep-DP round with the band kernels must return exactly
the hit set of the full-table kernels."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megapath_amd import hipapi as H  # noqa: E402
from megapath_amd import index_builder, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--pairs", type=int, default=4000)
args = ap.parse_args()
os.environ["MP_KMER_FILTER"] = "1"
os.environ["MP_LKT16"] = "1"
os.environ["MP_SA_DENSE"] = "1"
bases = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0
for rnd in range(args.rounds):
    rng = np.random.Generator(np.random.PCG64(9000 + rnd))
    genome = synth.random_genome(400000, 100 + rnd, n_seqs=3, repeat_frac=0.15)
    # tandem repeats: a unit of 7..45 bases copied 2..5 times in place
    g2 = []
    for name, arr in genome:
        arr = arr.copy()
        for _ in range(400):
            u = int(rng.integers(7, 46)); k = int(rng.integers(2, 6)); p = int(rng.integers(0, len(arr) - u * (k + 1)))
            for c in range(1, k + 1):
                arr[p + c * u:p + (c + 1) * u] = arr[p:p + u]
        g2.append((name, arr))
    genome = g2
    read_len = int(rng.choice([100, 125, 150, 150, 200, 250]))
    sim = dict(sub=float(rng.choice([0.003, 0.01, 0.02, 0.04, 0.06])), ins=float(rng.choice([0.0005, 0.002, 0.005])),
               dele=float(rng.choice([0.0005, 0.002, 0.005])), random_frac=0.02, broken_mate_frac=0.05,
               long_indel_frac=float(rng.choice([0.0, 0.2, 0.5])), long_indel_max=int(rng.choice([6, 15, 26])),
               junk_end_frac=float(rng.choice([0.0, 0.3])), junk_end_max=45, len_range=(read_len * 3 // 5, read_len),
               frag_mean=read_len + 200, frag_max=read_len + 450)
    names, r1, r2 = synth.simulate_pairs(genome, args.pairs, read_len, 500 + rnd, **sim)
    gix = H.Index.from_arrays(0, **index_builder.build_index_arrays(genome, sa_interval=16))
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, read_len + 1)
    res = {}
    for band in (False, True):
        for counts in (0, 1):
            ctx = H.Context(gix)
            ctx.set_reads(packed, lens)
            ctx.set_dp_band(band)
            hits, nh, st = ctx.deep_dp_round(H.AlignParams(1, 750, 130, 130, H.MmpParams.nt2(), counts, 0))
            res[(band, counts)] = sorted((h.pair_id, h.left_is_mate, h.algnmt_1, h.algnmt_2, h.score_1, h.score_2, h.start_1, h.start_2, h.dplen_1,
                                          h.dplen_2, h.count_1, h.count_2) for h in hits[:nh])
            stats = ctx.dp_band_stats()
            ctx.close()
    gix.close()
    strip = lambda rows: [r[:10] for r in rows]         # without tie counts: the full-table text-mode kernel does not produce them
    ok0 = strip(res[(True, 0)]) == strip(res[(False, 0)])
    ok1 = res[(True, 1)] == res[(False, 1)]
    ok = ok0 and ok1
    if not ok:
        for key, a, b in (("text", strip(res[(True, 0)]), strip(res[(False, 0)])), ("counts", res[(True, 1)], res[(False, 1)])):
            d = [(x, y) for x, y in zip(a, b) if x != y][:3]
            print("   %s: %d vs %d rows, first differences %s" % (key, len(a), len(b), d), flush=True)
    bad += 0 if ok else 1
    print("round %d: L=%d sub=%.3f indel=%.4f/%.4f long=%.1f/%d junk=%.1f: %d hits, band decided %d of %d -> %s" % (
        rnd, read_len, sim["sub"], sim["ins"], sim["dele"], sim["long_indel_frac"], sim["long_indel_max"], sim["junk_end_frac"], len(res[(True, 1)]),
        stats[0] - stats[1], stats[0], "identical" if ok else "DIFFERENT"), flush=True)
print("FAILED rounds: %d" % bad)
sys.exit(1 if bad else 0)
