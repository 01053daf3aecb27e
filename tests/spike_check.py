"""Checker for the device spike filter (row f4): the file pipeline of runMegaPath.sh:211-218 restated on numpy arrays.

bedtools is a third-party binary the reference calls but does not ship in a form that runs here, so its three steps are restated
from their documented behaviour (bamtobed: 0-based start, end = start + reference span of the CIGAR, name + /1 or /2;
genomecov -bga -ibam: depth per base from whole alignment spans, runs of equal depth INCLUDING zero-depth stretches; annotate:
fraction of each BED record covered by the union of the filter's intervals, printed with six decimals) -- that part of the
parity is unpinned.  cc/genomeCovFilter IS the reference's own code: it is restated here (genome_cov_filter) and pinned against the
real tool built by oracle/Makefile.ref (tests/test_host_logic.py) and used directly when the binary travelled with the snapshot."""
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "oracle", "_ref", "genomeCovFilter")


def bed_from_bam_text(lines):
    """records of tests/bamtext.py output (name flag chrom pos mapq cigar ...) -> [(chrom, start0, end0, name/mate)]"""
    out = []
    for ln in lines:
        f = ln.split("\t")
        name, flag, chrom, pos, cigar = f[0], int(f[1]), f[2], int(f[3]), f[5]
        if flag & 4 or chrom == "*":
            continue
        span, num = 0, 0
        for ch in cigar:
            if ch.isdigit():
                num = num * 10 + int(ch)
            else:
                if ch in "MDN=X":
                    span += num
                num = 0
        out.append((chrom, pos - 1, pos - 1 + span, name + ("/1" if flag & 64 else "/2" if flag & 128 else "")))
    return out


def bedgraph_bga(genome, bed):
    """genome: [(name, length)] in BAM header order -> [(chrom, from, to, depth)] as genomecov -bga lists them"""
    out = []
    by = {}
    for c, a, b, _ in bed:
        by.setdefault(c, []).append((a, b))
    for name, ln in genome:
        d = np.zeros(ln + 1, dtype=np.int64)
        for a, b in by.get(name, []):
            d[max(a, 0)] += 1
            d[min(b, ln)] -= 1
        depth = np.cumsum(d)[:ln]
        if ln == 0:
            continue
        cut = np.flatnonzero(np.diff(depth)) + 1
        starts = np.concatenate([[0], cut]); ends = np.concatenate([cut, [ln]])
        for a, b in zip(starts, ends):
            out.append((name, int(a), int(b), int(depth[a])))
    return out


def genome_cov_filter(genome, bedgraph, k: int):
    """cc/genomeCovFilter.cpp:58-92, operation for operation (running mean / sum of squared differences weighted by interval length)."""
    avg = {n: 0.0 for n, _ in genome}; diffp = {n: 0.0 for n, _ in genome}; cnt = {n: 0 for n, _ in genome}; var = {n: 0.0 for n, _ in genome}
    for c, a, b, d in bedgraph:
        if c not in avg:
            continue
        w = b - a
        delta = float(d) - avg[c]
        new_avg = avg[c] + delta * w / (cnt[c] + w)
        diffp[c] = diffp[c] + math.pow(delta, 2.0) * w * cnt[c] / (cnt[c] + w)
        var[c] = diffp[c] / (cnt[c] + w)
        cnt[c] += w
        avg[c] = new_avg
    mx = {c: avg[c] + k * math.sqrt(var[c]) for c in avg}
    return [(c, a, b) for c, a, b, d in bedgraph if c in mx and d > mx[c]], mx


def real_genome_cov_filter(genome, bedgraph, k: int, tmpdir):
    g, cov = os.path.join(tmpdir, "x.genome"), os.path.join(tmpdir, "x.genomecov")
    with open(g, "w") as f:
        for n, ln in genome:
            f.write("%s\t%d\n" % (n, ln))
    with open(cov, "w") as f:
        for c, a, b, d in bedgraph:
            f.write("%s\t%d\t%d\t%d\n" % (c, a, b, d))
    out = subprocess.run([REAL, g, cov, str(k)], stdout=subprocess.PIPE, check=True).stdout.decode()
    return [(l.split("\t")[0], int(l.split("\t")[1]), int(l.split("\t")[2])) for l in out.splitlines() if l]


def union(intervals):
    out = []
    for c, a, b in sorted(intervals):
        if out and out[-1][0] == c and out[-1][2] >= a:
            out[-1] = (c, out[-1][1], max(out[-1][2], b))
        else:
            out.append((c, a, b))
    return out


def read_filter(bed, spikes, threshold: float):
    """bedtools annotate + the awk of runMegaPath.sh:218 -> names to drop"""
    u = union(spikes)
    names = []
    for c, a, b, name in bed:
        cov = sum(max(0, min(b, y) - max(a, x)) for cc, x, y in u if cc == c)
        frac = float("%f" % (cov / (b - a))) if b > a else 0.0
        if frac >= threshold:
            names.append(name)
    return names
