#!/usr/bin/env python3
"""tests/golden/g5: the two stages of runMegaPath.sh that run with soap4.ini (mmpSeedMinLength 22, mmpReseedLen 23) instead of
soap4-nt2.ini -- stdout of the REAL soap4 (oracle/_ref) for the command lines the pipeline uses:

  h   host filter  (runMegaPath.sh:136)   -C soap4.ini -L 150 -T 1 -u 750 -F -nc              (150 bp reads: -L 150 cuts them to 149)
  r   16S filter   (runMegaPath.sh:162)   -C soap4.ini -L 150 -T 1 -u 750 -F -P -nc -top 100  (-P after -F: paired-only mode)
  n   the same reads with soap4-nt2.ini (-L 150 -F -nc): kept to show the read set tells the two ini files apart

The reads are drawn from g1's genome with 5-9 % substitutions in two thirds of the pairs: their longest exact matches are 15-30
bases, so a seed threshold of 22 instead of 17 changes which pairs are seeded, paired and rescued.
Build container only (needs /root/reference).  Data, not source."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from megapath_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
INI_HG = "/root/reference/soap4/soap4.ini"
INI_NT = "/root/reference/soap4/soap4-nt2.ini"


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g5 = os.path.join(HERE, "g5")
    os.makedirs(g5, exist_ok=True)
    genome = synth.random_genome(60000, 101, n_seqs=3, repeat_frac=0.08)          # = g1's genome (same generator call)
    tmp = tempfile.mkdtemp(prefix="g5_")
    try:
        fa = os.path.join(tmp, "ref.fa")
        synth.write_fasta(fa, genome)
        assert open(fa, "rb").read() == open(os.path.join(HERE, "g1", "ref.fa"), "rb").read()
        subprocess.check_call([os.path.join(REF, "2bwt-builder"), fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prefix = fa + ".index"
        names, r1, r2 = [], [], []
        for k, (n, sub, extra) in enumerate([(100, 0.01, dict(random_frac=0.05, broken_mate_frac=0.1)),
                                              (120, 0.06, dict(broken_mate_frac=0.1, ins=0.002, dele=0.002)),
                                              (120, 0.09, dict(noisy_mate_frac=0.2, noisy_sub=0.14))]):
            a, b, c = synth.simulate_pairs(genome, n, 150, seed=5100 + k, sub=sub, **extra)
            names += ["%s_%d" % (x, k) for x in a]; r1 += b; r2 += c
        f1, f2 = os.path.join(g5, "reads_1.fq"), os.path.join(g5, "reads_2.fq")
        synth.write_fastq_pair(f1, f2, names, r1, r2)
        runs = {"h": (INI_HG, ["-F", "-nc"]), "r": (INI_HG, ["-F", "-P", "-nc", "-top", "100"]), "n": (INI_NT, ["-F", "-nc"])}
        for tag, (ini, flags) in runs.items():
            cmd = [os.path.join(REF, "soap4"), "pair", prefix, f1, f2, "-o", os.path.join(tmp, "o"), "-C", ini, "-L", "150", "-T", "1", "-u", "750"] + flags
            with open(os.path.join(g5, "expected_%s.fq" % tag), "wb") as fo, open(os.path.join(tmp, "log"), "wb") as fe:
                subprocess.check_call(cmd, stdout=fo, stderr=fe)
        hs = [l for l in open(os.path.join(g5, "expected_h.fq")) if l.startswith("@")]
        ns = [l for l in open(os.path.join(g5, "expected_n.fq")) if l.startswith("@")]
        rs = [l for l in open(os.path.join(g5, "expected_r.fq")) if l.startswith("@")]
        print("records", len(hs), "headers differing between soap4.ini and soap4-nt2.ini:", sum(a != b for a, b in zip(sorted(hs), sorted(ns))),
              "| aligned h/n/r:", sum("SCORE:0;" not in l for l in hs), sum("SCORE:0;" not in l for l in ns), sum("SCORE:0;" not in l for l in rs))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
