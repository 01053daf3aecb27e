#!/usr/bin/env python3
"""Regenerates tests/golden/g2 from the REAL reference built by oracle/Makefile.ref (build container only: needs
/root/reference).  Everything written here is data -- inputs and the reference's outputs -- never source.

g2 pins two corners no g1 read reaches:

  set "t"  reads with MORE THAN 200 single-end candidates of EQUAL seed length.  singleMerge orders a read's candidates with
           std::sort(.., LongerSeedLength()) (DV-DPfunctions.cpp:334) and transferSeed keeps the first 200
           (DV-DPForSingleReads.cpp:200): which of the tied candidates survive depends on the permutation libstdc++'s
           introsort leaves them in.  Each family is an element R (9..11 blocks of 27 bases) present once per reference
           sequence; a copy carries ONE block exactly and two substitutions in every other block (8 and 18 bases into
           the block), so that its only seed is that block plus 8 bases either side (43 bases; 35 / 42 for the outer
           blocks), shared by the 31 (or 40) copies with the same exact block -- all within mmpSeedSAsizeThreshold + 1
           located rows.  The read is R itself (either strand), its mate is random: the pair fails the deep-DP round and
           the read goes through single-end DP with 279..341 candidates, 217..279 of them tied at 43.  Every copy scores
           the same, each on its own reference sequence, so the list of reference names in the output header IS the set
           of candidates that survived the cut.
  set "s"  pairs with reads of 1..16 bases (adapter-trimmed input; too short for a seed): the reference neither aborts
           nor aligns them, and their mates still go through single-end DP.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from megapath_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
INI = "/root/reference/soap4/soap4-nt2.ini"
B = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(rng, n):
    return B[rng.integers(0, 4, size=n)]


def other(rng, ch):
    return B[(int(np.searchsorted(B, ch)) + 1 + int(rng.integers(0, 3))) & 3]


def family(rng, tag, n_blocks, tail, copies_per_block, flank, sub_at=(8, 18)):
    """-> (consensus R, [(name, sequence)] one per copy)"""
    L = 27 * n_blocks + tail
    R = rnd(rng, L)
    seqs = []
    k = 0
    for b in range(n_blocks):
        for _ in range(copies_per_block):
            c = R.copy()
            for ob in range(n_blocks):
                if ob == b:
                    continue
                for off in sub_at:
                    c[27 * ob + off] = other(rng, R[27 * ob + off])
            # the bases next to the copy must not extend a match by chance: irrelevant here (the read ends with R)
            seqs.append(("%s_b%d_c%d" % (tag, b, k), np.concatenate([rnd(rng, flank), c, rnd(rng, flank)])))
            k += 1
    return R, seqs


def run_ref(prefix, fq1, fq2, L, out, extra=("-F", "-nc")):
    cmd = [os.path.join(REF, "soap4"), "pair", prefix, fq1, fq2, "-o", out + ".tmp", "-C", INI, "-L", str(L), "-T", "1", "-u", "750", *extra]
    with open(out, "wb") as fo, open(out + ".log", "wb") as fe:
        subprocess.check_call(cmd, stdout=fo, stderr=fe)


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g2 = os.path.join(HERE, "g2")
    os.makedirs(g2, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(2002))
    fams = [
        # tag, blocks, tail, copies per block, flank, substitution offsets inside a block
        ("fa", 9, 7, 31, 70, (8, 18)),      # 250 bp: 279 candidates, 217 tied at 43
        ("fb", 11, 3, 31, 70, (8, 18)),     # 300 bp: 341 candidates, 279 tied at 43
        ("fc", 9, 7, 40, 70, (8, 18)),      # 250 bp, 40 copies per exact block: only 31 SA rows of each seed are located
        ("fd", 9, 7, 31, 70, (7, 19)),      # 45-base seeds in the middle
        ("fe", 8, 9, 26, 70, (8, 18)),      # 225 bp: 208 candidates, 156 tied: the cut falls into the shorter classes
    ]
    genome, reads = [], []
    for tag, nb, tail, cpb, flank, sub_at in fams:
        R, seqs = family(rng, tag, nb, tail, cpb, flank, sub_at)
        genome += seqs
        reads.append((tag, R))
    # plain unique sequences for the ordinary and the short pairs
    uniq = [("u%d" % (i + 1), rnd(rng, 3000)) for i in range(4)]
    genome += uniq
    order = rng.permutation(len(genome))            # copies of one exact block are not neighbours in the text
    genome = [genome[i] for i in order]
    fa = os.path.join(g2, "ref.fa")
    synth.write_fasta(fa, genome)
    subprocess.check_call([os.path.join(REF, "2bwt-builder"), fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    prefix = fa + ".index"

    # ---- set "t": tied candidates beyond the cap of 200
    names, r1, r2 = [], [], []
    for k, (tag, R) in enumerate(reads):
        for strand in (0, 1):
            rd = R if strand == 0 else synth._COMP[R][::-1]
            mate = rnd(rng, len(R))
            names.append("%s_%s" % (tag, "fw" if strand == 0 else "rc"))
            if (k + strand) & 1:
                r1.append(mate.tobytes()); r2.append(rd.tobytes())
            else:
                r1.append(rd.tobytes()); r2.append(mate.tobytes())
    # a few ordinary pairs around them
    n2, a2, b2 = synth.simulate_pairs(uniq, n_pairs=12, read_len=150, seed=2003, sub=0.02, broken_mate_frac=0.3, len_range=(100, 300), frag_mean=420, frag_sd=40, frag_max=700)
    names += ["p" + x for x in n2]; r1 += a2; r2 += b2
    t1, t2 = os.path.join(g2, "reads_t_1.fq"), os.path.join(g2, "reads_t_2.fq")
    synth.write_fastq_pair(t1, t2, names, r1, r2)
    run_ref(prefix, t1, t2, 301, os.path.join(g2, "expected_t.fq"))

    # ---- set "s": reads too short to be seeded (< mmpSeedMinLength = 17 bases, e.g. what adapter trimming leaves over).
    # Reads of 17..29 bases are a different matter: they can get a DP instance whose cutoff (30) exceeds the read length,
    # for which callDP gives up on the WHOLE SIMD group of 32 (AVX2) or 16 (SSE4.1) instances the read happens to share
    # (CPU_DP.cpp:294-299, return value ignored at :803-809) -- build- and batch-order-dependent damage to unrelated
    # reads that nothing here reproduces (DESIGN.md "Known limits"); the product prints such reads unaligned and leaves
    # their neighbours alone, which tests/test_gpu_soap4_cli.py checks against the oracle.
    n3, a3, b3 = synth.simulate_pairs(uniq, n_pairs=40, read_len=150, seed=2004, sub=0.01, len_range=(30, 150), frag_mean=330, frag_sd=30)
    for k in range(len(n3)):
        if k % 3 == 0:
            a3[k] = a3[k][:int(rng.integers(1, 17))]
        if k % 5 == 0:
            b3[k] = b3[k][:int(rng.integers(1, 17))]
    s1, s2 = os.path.join(g2, "reads_s_1.fq"), os.path.join(g2, "reads_s_2.fq")
    synth.write_fastq_pair(s1, s2, n3, a3, b3)
    run_ref(prefix, s1, s2, 151, os.path.join(g2, "expected_s.fq"))
    os.remove(prefix + ".lkt")                      # 512 MiB, a pure function of .pac (LTConstruct.c:44-105): rebuilt by oracle and product
    for f in os.listdir(g2):
        if ".tmp" in f or f.endswith(".log"):
            os.remove(os.path.join(g2, f))
    print("g2 written:", sorted(os.listdir(g2)))


if __name__ == "__main__":
    main()
