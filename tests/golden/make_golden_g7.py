#!/usr/bin/env python3
"""Regenerates the `-b` chain goldens from the REAL reference (build container only): the NT loop exactly as runMegaPath.sh:191-227 runs
it -- EVERY run gets `-b -o <same prefix>` (runMegaPath.sh:199), reads `<prefix>.nt.tmp_in_{1,2}.fq` (the de-interleaved stdout of the
run before, moved there), only run 0 gets -nc -- so every run overwrites the BAM side files of the run before and what remains after
the loop are the LAST chunk's BAM files and the last run's stdout.  Data only:
  g6/expected_chain_b.bam.txt    nine chunks (g6's), g6's reads    (the stdout is asserted equal to g6/expected_chain.fq, made without -b)
  g3/expected_chain_a_b.bam.txt  three chunks (g3's), g1's read set a  (stdout == g3/expected_chain_a.fq)
The BAM files are decoded with tests/bamtext.py into one line per record (sections #header #dpout #gout #unpair), as for g1."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamtext  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
INI = "/root/reference/soap4/soap4-nt2.ini"


def deinterleave(src, out1, out2):
    lines = open(src, "rb").read().split(b"\n")
    with open(out1, "wb") as a, open(out2, "wb") as b:
        for i in range(0, len(lines) - 7, 8):
            a.write(b"\n".join(lines[i:i + 4]) + b"\n")
            b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")


def chain(work, fastas, reads1, reads2, L):
    """the loop of runMegaPath.sh:171-227 in `work`; returns (stdout of the last run, {file kind: decoded BAM})"""
    prefixes = []
    for k, fa in enumerate(fastas):
        dst = os.path.join(work, "db.%d" % k)
        shutil.copy(fa, dst)
        subprocess.check_call([os.path.join(REF, "2bwt-builder"), dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)     # (the .lkt is not kept in the repo)
        prefixes.append("db.%d.index" % k)
    shutil.copy(reads1, os.path.join(work, "S.nt.tmp_out.pe_1.fq"))
    shutil.copy(reads2, os.path.join(work, "S.nt.tmp_out.pe_2.fq"))
    last = None
    for i, prefix in enumerate(prefixes):
        for m in (1, 2):
            os.replace(os.path.join(work, "S.nt.tmp_out.pe_%d.fq" % m), os.path.join(work, "S.nt.tmp_in_%d.fq" % m))
        cmd = [os.path.join(REF, "soap4"), "pair", prefix, "S.nt.tmp_in_1.fq", "S.nt.tmp_in_2.fq", "-b", "-o", "S.nt.bam", "-L", str(L), "-T", "1", "-u", "750", "-F", "-C", INI, "-top", "95"]
        if i == 0:
            cmd.append("-nc")
        last = os.path.join(work, "stdout_%d" % i)
        with open(last, "wb") as fo, open(os.devnull, "wb") as fe:
            subprocess.check_call(cmd, cwd=work, stdout=fo, stderr=fe)
        deinterleave(last, os.path.join(work, "S.nt.tmp_out.pe_1.fq"), os.path.join(work, "S.nt.tmp_out.pe_2.fq"))
        os.remove(os.path.join(work, prefix + ".lkt"))
    text, _refs, dp = bamtext.read_bam(os.path.join(work, "S.nt.bam.dpout.1"))
    gout = bamtext.read_bam(os.path.join(work, "S.nt.bam.gout.1"))[2]
    unpair = bamtext.read_bam(os.path.join(work, "S.nt.bam.unpair"))[2]
    return open(last, "rb").read(), text, dp, gout, unpair


def write(out_bam, res):
    _stdout, text, dp, gout, unpair = res
    with open(out_bam, "w") as f:
        f.write("#header\n" + text)
        for tag, recs in (("dpout", dp), ("gout", gout), ("unpair", unpair)):
            f.write("#%s\n" % tag + "".join(r + "\n" for r in recs))
    print(os.path.basename(out_bam), len(dp), len(gout), len(unpair))


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    g1, g3, g6 = (os.path.join(HERE, d) for d in ("g1", "g3", "g6"))
    with tempfile.TemporaryDirectory(dir="/tmp") as work:
        res = chain(work, [os.path.join(g6, "c%d.fa" % k) for k in range(9)], os.path.join(g6, "reads_1.fq"), os.path.join(g6, "reads_2.fq"), 151)
        # -b must not change what the chain prints (g6/expected_chain.fq was made without it)
        assert res[0] == open(os.path.join(g6, "expected_chain.fq"), "rb").read()
        write(os.path.join(g6, "expected_chain_b.bam.txt"), res)
    with tempfile.TemporaryDirectory(dir="/tmp") as work:
        res = chain(work, [os.path.join(g3, "c%d.fa" % k) for k in range(3)], os.path.join(g1, "reads_a_1.fq"), os.path.join(g1, "reads_a_2.fq"), 151)
        assert res[0] == open(os.path.join(g3, "expected_chain_a.fq"), "rb").read()
        write(os.path.join(g3, "expected_chain_a_b.bam.txt"), res)


if __name__ == "__main__":
    main()
