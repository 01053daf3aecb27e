#!/usr/bin/env python3
"""Regenerates tests/golden/g3 from the REAL reference (build container only): the NT loop of runMegaPath.sh:191-227 over a
database cut into three index chunks.  Data only: three FASTA chunks (the three sequences of g1/ref.fa), their index files as
the reference's 2bwt-builder writes them, and the stdout of the last of three chained `soap4 pair` runs -- run k+1 reads the
de-interleaved stdout of run k and merges the SCORE: comments (BGS-IO.cpp:1348-1371); only run 0 gets -nc (runMegaPath.sh:206-208).
The reads are g1's sets "a" (2x150) and "c" (75-300 bp, 30 % broken mates)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
INI = "/root/reference/soap4/soap4-nt2.ini"


def deinterleave(src, out1, out2):
    lines = open(src, "rb").read().split(b"\n")
    with open(out1, "wb") as a, open(out2, "wb") as b:
        for i in range(0, len(lines) - 7, 8):
            a.write(b"\n".join(lines[i:i + 4]) + b"\n")
            b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g1, g3 = os.path.join(HERE, "g1"), os.path.join(HERE, "g3")
    os.makedirs(g3, exist_ok=True)
    recs = open(os.path.join(g1, "ref.fa"), "rb").read().split(b">")[1:]
    assert len(recs) == 3
    prefixes = []
    for k, rec in enumerate(recs):
        fa = os.path.join(g3, "c%d.fa" % k)
        open(fa, "wb").write(b">" + rec)
        subprocess.check_call([os.path.join(REF, "2bwt-builder"), fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prefixes.append(fa + ".index")
    for tag, L in (("a", 151), ("c", 301)):
        in1, in2 = os.path.join(g1, "reads_%s_1.fq" % tag), os.path.join(g1, "reads_%s_2.fq" % tag)
        for k, prefix in enumerate(prefixes):
            out = os.path.join(g3, "chain_%s_%d.tmp" % (tag, k))
            cmd = [os.path.join(REF, "soap4"), "pair", prefix, in1, in2, "-o", os.path.join(g3, "o.tmp"), "-C", INI, "-L", str(L), "-T", "1", "-u", "750", "-F", "-top", "95"]
            if k == 0:
                cmd.append("-nc")
            with open(out, "wb") as fo, open(os.devnull, "wb") as fe:
                subprocess.check_call(cmd, stdout=fo, stderr=fe)
            in1, in2 = out + ".1", out + ".2"
            deinterleave(out, in1, in2)
        os.replace(out, os.path.join(g3, "expected_chain_%s.fq" % tag))
    for f in os.listdir(g3):
        if ".tmp" in f or f.endswith(".lkt"):
            os.remove(os.path.join(g3, f))
    print("g3 written:", sorted(os.listdir(g3)))


if __name__ == "__main__":
    main()
