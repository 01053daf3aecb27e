#!/usr/bin/env python3
"""Regenerates the BAM goldens of tests/golden/g1 from the REAL reference built by oracle/Makefile.ref (build container
only: needs /root/reference).  Data only: inputs and the reference's outputs.

The reference `soap4 ... -b` writes <prefix>.dpout.1 (deep-DP pairs), <prefix>.gout.N (mate-rescued pairs) and
<prefix>.unpair; they are decoded with tests/bamtext.py and kept as text, one line per BAM record:
  expected_<set>.bam.txt   sections "#header", "#dpout", "#gout", "#unpair"
Sets: a (-F -nc), c (mixed lengths, -L 301), d (-P with IGNORE comments), g (noisy mates -> mate rescue, random
qualities, -p for NM/MD), h (fragments straddling sequence boundaries -> BoundaryCheckDP trimming).  Sets g and h also
get reads_<set>_{1,2}.fq and expected_<set>.fq (stdout).  All runs use -T 1 and relative file names from inside the
index directory, so that the read group (= first read file as given) is "reads_<set>_1.fq".
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from megapath_amd import synth  # noqa: E402
import bamtext  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
INI = "/root/reference/soap4/soap4-nt2.ini"
G1 = os.path.join(HERE, "g1")


def write_fastq_random_qual(path1, path2, names, r1, r2, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    with open(path1, "wb") as f1, open(path2, "wb") as f2:
        for n, a, b in zip(names, r1, r2):
            qa = bytes((33 + rng.integers(2, 41, size=len(a))).astype(np.uint8))
            qb = bytes((33 + rng.integers(2, 41, size=len(b))).astype(np.uint8))
            f1.write(b"@" + n.encode() + b"/1\n" + a + b"\n+\n" + qa + b"\n")
            f2.write(b"@" + n.encode() + b"/2\n" + b + b"\n+\n" + qb + b"\n")


def boundary_pairs(genome, n_pairs, seed):
    """Fragments of the concatenated text placed so that one mate overlaps a sequence junction."""
    rng = np.random.Generator(np.random.PCG64(seed))
    text = np.concatenate([s[1] for s in genome])
    cuts = np.cumsum([len(s[1]) for s in genome])[:-1]
    comp = synth._COMP
    names, r1, r2 = [], [], []
    for p in range(n_pairs):
        cut = int(cuts[p % len(cuts)])
        flen = int(rng.integers(300, 500))
        if p % 2 == 0:
            start = cut - int(rng.integers(10, 141))              # first mate crosses the junction
        else:
            start = cut - flen + int(rng.integers(10, 141))       # second mate crosses it
        start = max(0, min(start, len(text) - flen))
        frag = synth._mutate(text[start:start + flen], rng, 0.01, 0.001, 0.001)
        a = frag[:150]
        b = comp[frag[len(frag) - 150:]][::-1]
        if rng.random() < 0.5:
            a, b = b, a
        names.append("h%d" % (p + 1))
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return names, r1, r2


def main():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "-j8"], cwd=ROOT)
    genome = synth.random_genome(60000, 101, n_seqs=3, repeat_frac=0.08)
    # new read sets (kept in the repo)
    names, r1, r2 = synth.simulate_pairs(genome, n_pairs=200, read_len=150, seed=105, sub=0.02, ins=0.002, dele=0.002,
                                         random_frac=0.03, broken_mate_frac=0.05, noisy_mate_frac=0.4, noisy_sub=0.2, n_frac=0.001)
    write_fastq_random_qual(os.path.join(G1, "reads_g_1.fq"), os.path.join(G1, "reads_g_2.fq"), names, r1, r2, 205)
    names, r1, r2 = boundary_pairs(genome, 80, 106)
    write_fastq_random_qual(os.path.join(G1, "reads_h_1.fq"), os.path.join(G1, "reads_h_2.fq"), names, r1, r2, 206)

    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        work = os.path.join(tmp, "g1")
        shutil.copytree(G1, work)
        # the committed index has no .lkt (512 MiB, a pure function of .pac): let the real builder write it again
        subprocess.check_call([os.path.join(REF, "2bwt-builder"), "ref.fa"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for ext in ("bwt", "sa", "pac", "tra"):      # (.ann carries a time-based seed in its first line)
            assert open(os.path.join(work, "ref.fa.index." + ext), "rb").read() == open(os.path.join(G1, "ref.fa.index." + ext), "rb").read(), ext
        runs = {
            "a": (["-L", "151", "-F", "-nc"], True),
            "c": (["-L", "301", "-F", "-nc"], True),
            "d": (["-L", "151", "-P"], True),
            "g": (["-L", "151", "-F", "-p"], False),
            "h": (["-L", "151", "-F"], False),
            # `-b` alone (no -F / -P): nothing on stdout, and the BAM records are made from the results as the DP left them (the
            # MegaPath modes rewrite them first: BGS-IO.cpp:2103-2110 -> :1984-2011) -- MAPQ / X0 / X1 differ for some pairs
            "a.bonly": (["-L", "151"], None),
            "g.bonly": (["-L", "151", "-p"], None),
            "h.bonly": (["-L", "151"], None),
        }
        for k, (flags, have_fq) in runs.items():
            rk = k.split(".")[0]
            cmd = [os.path.join(REF, "soap4"), "pair", "ref.fa.index", "reads_%s_1.fq" % rk, "reads_%s_2.fq" % rk, "-o", "out_" + k,
                   "-C", INI, "-T", "1", "-u", "750", "-b"] + flags
            with open(os.path.join(work, "stdout_%s.fq" % k), "wb") as fo, open(os.devnull, "wb") as fe:
                subprocess.check_call(cmd, cwd=work, stdout=fo, stderr=fe)
            got = open(os.path.join(work, "stdout_%s.fq" % k), "rb").read()
            if have_fq is None:
                assert got == b"", k
            elif have_fq:       # -b must not change the annotated FASTQ already pinned
                assert got == open(os.path.join(G1, "expected_%s.fq" % k), "rb").read(), k
            else:
                open(os.path.join(G1, "expected_%s.fq" % k), "wb").write(got)
            text, _refs, dp = bamtext.read_bam(os.path.join(work, "out_%s.dpout.1" % k))
            gout = bamtext.read_bam(os.path.join(work, "out_%s.gout.1" % k))[2]
            unpair = bamtext.read_bam(os.path.join(work, "out_%s.unpair" % k))[2]
            with open(os.path.join(G1, "expected_%s.bam.txt" % k), "w") as f:
                f.write("#header\n" + text)
                for tag, recs in (("dpout", dp), ("gout", gout), ("unpair", unpair)):
                    f.write("#%s\n" % tag + "".join(r + "\n" for r in recs))
            print(k, len(dp), len(gout), len(unpair))


if __name__ == "__main__":
    main()
