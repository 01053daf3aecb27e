#!/usr/bin/env python3
"""Regenerates tests/golden/g1/expected_*.lsam with the REAL cc/fastq2lsam (built by oracle/Makefile.ref from the sources
under /root/reference; build container only).  Data only: the reference tool's stdout for the committed expected_*.fq."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
G1 = os.path.join(HERE, "g1")


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "oracle/_ref/fastq2lsam"], cwd=ROOT)
    tool = os.path.join(ROOT, "oracle", "_ref", "fastq2lsam")
    for tag, arg in (("a", "1"), ("d", "1"), ("e", "1"), ("g", "1"), ("h", "1"), ("a.noseq", "0")):
        src = os.path.join(G1, "expected_%s.fq" % tag.split(".")[0])
        with open(src, "rb") as fi, open(os.path.join(G1, "expected_%s.lsam" % tag), "wb") as fo:
            subprocess.check_call([tool, arg], stdin=fi, stdout=fo)


if __name__ == "__main__":
    main()
