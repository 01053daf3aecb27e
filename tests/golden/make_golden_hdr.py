#!/usr/bin/env python3
"""tests/golden/hdr: FASTA headers the way NT writes them -- a '>' inside a description ("3'->5' exonuclease"), a header that
begins with white space, tabs, a last header without a newline (an empty record) -- and the index files the REAL 2bwt-builder
(oracle/_ref, built by oracle/Makefile.ref) wrote for it.  Build container only (needs /root/reference); data, not source."""
import os
import random
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    d = os.path.join(HERE, "hdr")
    os.makedirs(d, exist_ok=True)
    rng = random.Random(4242)
    seq = lambda n: "".join(rng.choice("ACGT") for _ in range(n))          # noqa: E731

    def wrap(s, w=70):
        return "\n".join(s[i:i + w] for i in range(0, len(s), w)) + "\n"
    fa = (">AB000001.1 Homo sapiens 3'->5' exonuclease domain (a > b) mRNA\n" + wrap(seq(1500))
          + ">CD000002.2\tpartial>cds >>> end\n" + wrap(seq(900))
          + "> leading_space_name description\n" + wrap(seq(700))
          + ">EF000003.1 plain\n" + wrap(seq(1100))
          + ">last_header_without_newline and > in it")
    path = os.path.join(d, "h.fa")
    with open(path, "w") as f:
        f.write(fa)
    subprocess.check_call([os.path.join(REF, "2bwt-builder"), path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for ext in (".lkt", ".rev.lkt", ".rev.bwt", ".rev.fmv", ".rev.pac", ".sai", ".bwt.occ", ".rev.bwt.occ"):
        p = path + ".index" + ext
        if os.path.exists(p):
            os.remove(p)
    print(sorted(os.listdir(d)))
    print(open(path + ".index.ann").read())


if __name__ == "__main__":
    main()
