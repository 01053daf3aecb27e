#!/usr/bin/env python3
"""Regenerates tests/golden/g6 from the REAL reference (build container only): the NT loop of runMegaPath.sh:191-227 over a database
cut into NINE index chunks with homologous sequences in several of them (copies of two ancestors at 0-7 % divergence), so that a read
collects hits of different scores from many chunks and the -top 95 rule of the comment merge (BGS-IO.cpp:1348-1371, 2034-2089) decides
chunk after chunk what stays in the header.  Data only: nine FASTA chunks, their index files as the reference's 2bwt-builder writes
them, the reads, and the stdout of the last of nine chained `soap4 pair` runs (run k+1 reads the de-interleaved stdout of run k; only
run 0 gets -nc, runMegaPath.sh:206-208)."""
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
N_CHUNKS = 9
B = np.frombuffer(b"ACGT", dtype=np.uint8)


def deinterleave(src, out1, out2):
    lines = open(src, "rb").read().split(b"\n")
    with open(out1, "wb") as a, open(out2, "wb") as b:
        for i in range(0, len(lines) - 7, 8):
            a.write(b"\n".join(lines[i:i + 4]) + b"\n")
            b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")


def mutate(rng, seq, div):
    s = seq.copy()
    k = rng.random(len(s)) < div
    s[k] = B[(np.searchsorted(B, s[k]) + rng.integers(1, 4, size=int(k.sum()))) & 3]
    return s


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g6 = os.path.join(HERE, "g6")
    os.makedirs(g6, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(6006))
    anc = [B[rng.integers(0, 4, size=6000)] for _ in range(2)]
    plan = {0: [(0, 0.0)], 1: [(1, 0.03)], 2: [(0, 0.02), (1, 0.0)], 3: [], 4: [(0, 0.07)], 5: [(0, 0.04)], 6: [(1, 0.06)], 7: [(0, 0.01), (1, 0.02)], 8: [(1, 0.045)]}
    chunks, prefixes = [], []
    for k in range(N_CHUNKS):
        seqs = [("c%d_uniq" % k, B[rng.integers(0, 4, size=5000)])]
        for a, div in plan[k]:
            seqs.append(("c%d_anc%d_d%d" % (k, a, int(div * 1000)), np.concatenate([B[rng.integers(0, 4, size=300)], mutate(rng, anc[a], div), B[rng.integers(0, 4, size=300)]])))
        chunks.append(seqs)
        fa = os.path.join(g6, "c%d.fa" % k)
        synth.write_fasta(fa, seqs)
        subprocess.check_call([os.path.join(REF, "2bwt-builder"), fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prefixes.append(fa + ".index")
    every = [s for c in chunks for s in c]
    names, r1, r2 = synth.simulate_pairs(every, 300, 150, seed=6100, sub=0.015, ins=0.001, dele=0.001, random_frac=0.08, broken_mate_frac=0.12)
    f1, f2 = os.path.join(g6, "reads_1.fq"), os.path.join(g6, "reads_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    in1, in2 = f1, f2
    for k, prefix in enumerate(prefixes):
        out = os.path.join(g6, "chain_%d.tmp" % k)
        cmd = [os.path.join(REF, "soap4"), "pair", prefix, in1, in2, "-o", os.path.join(g6, "o.tmp"), "-C", INI, "-L", "151", "-T", "1", "-u", "750", "-F", "-top", "95"]
        if k == 0:
            cmd.append("-nc")
        with open(out, "wb") as fo, open(os.devnull, "wb") as fe:
            subprocess.check_call(cmd, stdout=fo, stderr=fe)
        in1, in2 = out + ".1", out + ".2"
        deinterleave(out, in1, in2)
    os.replace(out, os.path.join(g6, "expected_chain.fq"))
    for f in os.listdir(g6):
        if ".tmp" in f or f.endswith(".lkt"):
            os.remove(os.path.join(g6, f))
    hdr = [l for l in open(os.path.join(g6, "expected_chain.fq")) if l.startswith("@")]
    multi = sum(1 for l in hdr if l.count(",c") >= 2)
    print("g6 written: %d records, %d with hits from two or more sequences, %d unaligned" % (len(hdr), multi, sum("SCORE:0;" in l for l in hdr)))


if __name__ == "__main__":
    main()
