#!/usr/bin/env python3
"""Regenerates tests/golden/g4 from the REAL reference tools (build container only): the tail of the NT stage,
    soap4 pair ... -F | cc/fastq2lsam 1 | cc/taxLookupAcc acc2tid nodes.dmp names.dmp - > LSAM.id ;  cc/reassign -t 40 LSAM.id
(runMegaPath.sh:208,247).  Data only: a small database of five species (some with several strain sequences, two of them close
relatives, one a near-subset of another), a synthetic NCBI-style taxonomy for it, reads, and the tools' outputs:
  expected.fq           soap4's stdout
  expected.lsam.id      after fastq2lsam + taxLookupAcc (species-level taxids, super-kingdom names appended)
  expected.explains     stderr of reassign: the "<t1> explains <t2>" lines its pass 1 derives from counts / uniq_counts / intersect
  expected.ra.lsam.id   reassign's stdout (pass 2)
Sequence names use the forms getCorrectAcc / removeVersion understand (cc/misc.h:29-45, cc/taxonomy.h:14-24): plain
`ACC.version` and old-style `gi|123|gb|ACC.version|`."""
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


def mutate(rng, seq, rate):
    out = seq.copy()
    hit = rng.random(len(seq)) < rate
    out[hit] = B[(np.searchsorted(B, out[hit]) + 1 + rng.integers(0, 3, size=int(hit.sum()))) & 3]
    return out


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g4 = os.path.join(HERE, "g4")
    os.makedirs(g4, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(4004))
    rnd = lambda n: B[rng.integers(0, 4, size=n)]  # noqa: E731
    a = rnd(24000)                                   # species 1001, strain 10011
    a2 = mutate(rng, a, 0.004)                       # species 1001, another strain (taxid 10012): same species after popUpToSpecies
    b = np.concatenate([mutate(rng, a[3000:9000], 0.01), rnd(6000)])     # species 1002: half of it a close relative of 1001
    c = np.concatenate([mutate(rng, a[12000:15000], 0.002), rnd(150)])   # species 1003: nearly a subset of 1001 -> "1001 explains 1003"
    d = rnd(15000)                                   # species 2001 (another super-kingdom), unrelated
    e = np.concatenate([d[2000:5000], rnd(9000)])    # species 2002: shares 3 kb with 2001 exactly
    genome = [("NC_000101.1", a), ("gi|555|gb|CP000102.2|", a2), ("NZ_ABCD01000103.1", b), ("AB000104.1", c), ("gi|777|ref|NC_000205.1|", d),
              ("NC_000206", e), ("XX_999999.1", rnd(4000))]       # the last accession is missing from acc2tid: "Taxid not found"
    fa = os.path.join(g4, "ref.fa")
    synth.write_fasta(fa, genome)
    # ---- taxonomy: 1 root; 2, 3 super-kingdoms; genus 10, 20; species 1001.. ; strains ("no rank") below two of the species
    nodes = [(1, 1, "no rank"), (2, 1, "superkingdom"), (3, 1, "superkingdom"), (10, 2, "genus"), (20, 3, "genus"), (1001, 10, "species"), (1002, 10, "species"),
             (1003, 10, "species"), (2001, 20, "species"), (2002, 20, "species"), (10011, 1001, "no rank"), (10012, 1001, "no rank"), (20021, 2002, "no rank")]
    names = {1: "root", 2: "Bacteria", 3: "Viruses", 10: "Alphagenus", 20: "Betavirus", 1001: "Alphagenus primus", 1002: "Alphagenus secundus", 1003: "Alphagenus tertius",
             2001: "Betavirus unus", 2002: "Betavirus duo", 10011: "Alphagenus primus str. K1", 10012: "Alphagenus primus str. K2", 20021: "Betavirus duo isolate 7"}
    with open(os.path.join(g4, "nodes.dmp"), "w") as f:
        for t, p, r in nodes:
            f.write("%d\t|\t%d\t|\t%s\t|\t\t|\t0\t|\n" % (t, p, r))
    with open(os.path.join(g4, "names.dmp"), "w") as f:
        for t, n in names.items():
            f.write("%d\t|\t%s\t|\t\t|\tscientific name\t|\n" % (t, n))
            f.write("%d\t|\t%s synonym\t|\t\t|\tsynonym\t|\n" % (t, n))
    acc = [("NC_000101", "NC_000101.1", 10011, 1), ("CP000102", "CP000102.2", 10012, 2), ("NZ_ABCD01000103", "NZ_ABCD01000103.1", 1002, 3), ("AB000104", "AB000104.1", 1003, 4),
           ("NC_000205", "NC_000205.1", 2001, 5), ("NC_000206", "NC_000206.1", 20021, 6), ("ZZ_000001", "ZZ_000001.1", 1002, 7)]
    with open(os.path.join(g4, "acc2tid.txt"), "w") as f:
        f.write("accession\taccession.version\ttaxid\tgi\n")
        for row in acc:
            f.write("%s\t%s\t%d\t%d\n" % row)
    subprocess.check_call([os.path.join(REF, "2bwt-builder"), fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    prefix = fa + ".index"
    # ---- reads: mostly from species 1001, some from the others, a few random; some with IGNORE comments
    pools = [([genome[0], genome[1]], 520), ([genome[2]], 60), ([genome[3]], 14), ([genome[4]], 90), ([genome[5]], 70), ([genome[6]], 12)]
    names_, r1, r2 = [], [], []
    for k, (seqs, n) in enumerate(pools):
        nn, a1, b1 = synth.simulate_pairs(seqs, n_pairs=n, read_len=150, seed=4100 + k, sub=0.01, random_frac=0.03, broken_mate_frac=0.05)
        names_ += ["q%d_%s" % (k, x) for x in nn]; r1 += a1; r2 += b1
    order = rng.permutation(len(names_))
    names_ = [names_[i] for i in order]; r1 = [r1[i] for i in order]; r2 = [r2[i] for i in order]
    f1, f2 = os.path.join(g4, "reads_1.fq"), os.path.join(g4, "reads_2.fq")
    synth.write_fastq_pair(f1, f2, names_, r1, r2)
    out_fq = os.path.join(g4, "expected.fq")
    cmd = [os.path.join(REF, "soap4"), "pair", prefix, f1, f2, "-o", os.path.join(g4, "o.tmp"), "-C", INI, "-L", "151", "-T", "1", "-u", "750", "-F", "-nc", "-top", "95"]
    with open(out_fq, "wb") as fo, open(os.devnull, "wb") as fe:
        subprocess.check_call(cmd, stdout=fo, stderr=fe)
    lsam = subprocess.run([os.path.join(REF, "fastq2lsam"), "1"], stdin=open(out_fq, "rb"), stdout=subprocess.PIPE, check=True).stdout
    p = subprocess.run([os.path.join(REF, "taxLookupAcc"), os.path.join(g4, "acc2tid.txt"), os.path.join(g4, "nodes.dmp"), os.path.join(g4, "names.dmp"), "-"],
                       input=lsam, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    open(os.path.join(g4, "expected.lsam.id"), "wb").write(p.stdout)
    r = subprocess.run([os.path.join(REF, "reassign"), "-t", "40", os.path.join(g4, "expected.lsam.id")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    open(os.path.join(g4, "expected.ra.lsam.id"), "wb").write(r.stdout)
    open(os.path.join(g4, "expected.explains"), "wb").write(b"".join(sorted(l + b"\n" for l in r.stderr.split(b"\n") if b"explains" in l)))
    for f in os.listdir(g4):
        if ".tmp" in f or f.endswith(".lkt"):
            os.remove(os.path.join(g4, f))
    print("g4 written; explains:", open(os.path.join(g4, "expected.explains")).read().split("\n")[:6])
    print("taxLookupAcc stderr:", sorted(set(p.stderr.decode().split("\n")))[:6])


if __name__ == "__main__":
    main()
