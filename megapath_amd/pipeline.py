"""Host side of the soap4 paired-end path above the C ABI (product code; never imports oracle/).

Mirrors `soap3_dp_pair_align` (reference soap4/alignment.cpp:29-355): the deep-DP round runs entirely on the GPU
(`mp_deep_dp_round`); pairs it leaves unaligned go through single-end DP on their pooled seeds
(DV-DPForSingleReads.cpp:121-215, DV-DPfunctions.cpp:295-342) and mate rescue inside the insert window
(DV-SemiDP.cpp, DV-DPfunctions.cpp:1048-1231) -- candidate selection, windows and both DP batches on the device
(`mp_rescue_round`) -- and everything is written as the annotated interleaved FASTQ of `-F` / `-P` mode (BGS-IO.cpp:1384-1446, 1966-2091).
Host work here is per-pair bookkeeping only; every DP cell and every FM-index step happens in the HIP kernels.
"""
from __future__ import annotations

import gzip
import io
from dataclasses import dataclass, field

import numpy as np

from . import hipapi as H

NOT_ALIGNED = (1 << 64) - 1


def dp_cutoff(length: int) -> int:
    return int(max(0.2 * length, 30.0))                   # definitions.h:166-167


def margin(length: int) -> int:
    return 30 if length > 100 else 25                     # DP2_MARGIN / DPS_MARGIN


@dataclass
class Soap4Params:
    max_read_length: int = 120                            # -L (IniParam.cpp:627-648 defaults)
    insert_low: int = 1                                   # -v
    insert_high: int = 500                                # -u
    clip_left: int = 130                                  # ini MaxFrontLenClipped
    clip_right: int = 130                                 # ini MaxEndLenClipped
    megapath_mode: int = 1                                # 1 = -F, 2 = -P
    top_percentage: float = 0.95                          # -top N
    skip_default_dp: bool = False
    ignore_comments: bool = False                         # -nc (SOAP4.cpp:537-547)
    mmp: H.MmpParams = field(default_factory=H.MmpParams.nt2)


@dataclass
class PathStats:
    n_pairs: int = 0
    n_deep_pairs: int = 0
    n_rescued_pairs: int = 0
    n_unpaired_pairs: int = 0
    n_dp_instances: int = 0


def _open(path):
    if path.endswith(".gz"):
        return io.BufferedReader(gzip.open(path, "rb"))
    return open(path, "rb")


def read_fastq(path):
    """Yield (name, comment, seq, qual); name loses a trailing /1 /2 (QueryParser.cpp:177), comment = rest of header."""
    with _open(path) as f:
        while True:
            h = f.readline()
            if not h:
                return
            h = h.rstrip(b"\r\n")
            if not h:
                continue
            seq = f.readline().rstrip(b"\r\n")
            fasta = h[:1] == b">"
            qual = b""
            if not fasta:
                f.readline()
                qual = f.readline().rstrip(b"\r\n")
            body = h[1:]
            cut = len(body)
            for i, ch in enumerate(body):
                if ch in (32, 9):
                    cut = i
                    break
            name, comment = body[:cut], body[cut + 1:]
            if len(name) > 2 and name[-2:-1] == b"/" and name[-1:].isdigit():
                name = name[:-2]
            yield name, (comment if comment else None), seq, qual


def _mapping_from_header(comment, top, score_t):
    """getMappingFromHeader (BGS-IO.cpp:1348-1371), quirks included."""
    v = []
    if comment is None or comment == b"IGNORE":
        return 0, v
    def atoi(b):
        n, i = 0, 0
        while i < len(b) and b[i:i + 1].isdigit():
            n = n * 10 + b[i] - 48
            i += 1
        return n
    score = atoi(comment[6:])
    if score < score_t:
        return score, v
    elif score_t < score * top:
        score_t = score * top
    p = comment.find(b";", 6)
    while p >= 0 and p + 1 < len(comment):
        s = p + 1
        p2 = comment.find(b";", s)
        if p2 < 0:
            break
        if score >= score_t:
            v.append((atoi(comment[s:]), comment[s:p2]))
        p = p2
    return score, v


class _Writer:
    def __init__(self, index, params, out):
        self.ix, self.p, self.out = index, params, out

    def _score_line(self, name, comment, best, hits, v):
        top = self.p.top_percentage
        if comment is not None and comment == b"IGNORE":
            return b"@" + name + b"\tIGNORE\n"
        s = b"@" + name + b"\tSCORE:%d;" % best
        if best > 0:
            prev = None
            for chr_id, neg in hits:
                if prev is not None and chr_id == prev:
                    continue
                prev = chr_id
                if -neg > 0 and -neg >= best * top:
                    s += b"%d,%s;" % (-neg, self.ix.seq_name(chr_id).encode())
        for sc, txt in v:
            if sc >= best * top:
                s += txt + b";"
        return s + b"\n"

    def pair(self, rec1, rec2, results, chr1, chr2):
        """pairDeepDPOutputFastqAPI (BGS-IO.cpp:1966-2091). results: list of [a1, a2, s1, s2]; chr*: per-result ids or -1."""
        best1 = best2 = 0
        h1, h2 = [], []
        for r, c1, c2 in zip(results, chr1, chr2):
            a1, a2, s1, s2 = r
            if c1 == -1:
                a1, s1 = NOT_ALIGNED, 0
            if c2 == -1:
                a2, s2 = NOT_ALIGNED, 0
            if self.p.megapath_mode == 2 and (a1 == NOT_ALIGNED or a2 == NOT_ALIGNED):
                a1 = a2 = NOT_ALIGNED
                s1 = s2 = 0
            if c1 == c2 and a1 != NOT_ALIGNED and a2 != NOT_ALIGNED:      # normalizeScore (:1943-1961)
                s1 = s2 = s1 + s2
            best1, best2 = max(best1, s1), max(best2, s2)
            if c1 != -1:
                h1.append((c1, -s1))
            if c2 != -1:
                h2.append((c2, -s2))
        h1.sort()
        h2.sort()
        top = self.p.top_percentage
        p1, v1 = _mapping_from_header(rec1[1], top, best1 * top)
        p2, v2 = _mapping_from_header(rec2[1], top, best2 * top)
        best1, best2 = max(best1, p1), max(best2, p2)
        self.out.write(self._score_line(rec1[0], rec1[1], best1, h1, v1) + rec1[2] + b"\n+\n" + rec1[3] + b"\n")
        self.out.write(self._score_line(rec2[0], rec2[1], best2, h2, v2) + rec2[2] + b"\n+\n" + rec2[3] + b"\n")

    def single(self, rec, hits_chr_score):
        """unproperlypairDPOutputFastqAPI (BGS-IO.cpp:1384-1446). hits_chr_score: list of (chr or -1, score)."""
        if self.p.megapath_mode == 2:
            hits_chr_score = []
        if rec[1] is not None and rec[1] == b"IGNORE":
            self.out.write(b"@" + rec[0] + b"\tIGNORE\n" + rec[2] + b"\n+\n" + rec[3] + b"\n")
            return
        best, h = 0, []
        for c, s in hits_chr_score:
            if c < 0:
                s = 0
            else:
                h.append((c, -s))
            best = max(best, s)
        h.sort()
        top = self.p.top_percentage
        prev, v = _mapping_from_header(rec[1], top, best * top)
        best = max(best, prev)
        self.out.write(self._score_line(rec[0], rec[1], best, h, v) + rec[2] + b"\n+\n" + rec[3] + b"\n")


def _target_chr(index, positions, read_lens):
    """decideTargetChr (BGS-IO.cpp:1312-1341) vectorised: sequence id when [pos, pos+len-1] stays in one sequence."""
    positions = np.asarray(positions, dtype=np.uint64)
    if len(positions) == 0:
        return np.zeros(0, dtype=np.int64)
    _, c1 = index.chr_and_pos_batch(positions)
    _, c2 = index.chr_and_pos_batch(positions + np.asarray(read_lens, dtype=np.uint64) - np.uint64(1))
    out = c1.astype(np.int64)
    out[c1 != c2] = -1
    return out


class PairAligner:
    """One GPU, one index, batches of pairs -> annotated FASTQ."""

    def __init__(self, index: H.Index, params: Soap4Params):
        self.ix, self.p = index, params
        self.ctx = H.Context(index)
        self.stats = PathStats()
        self._first_batch = True
        self.text_len = int(index.info.text_length)

    def close(self):
        self.ctx.close()

    # ------------------------------------------------------------------ one batch
    def align_batch(self, recs, out):
        """recs: list of (name, comment, seq, qual) with 2 entries per pair (read, mate)."""
        p = self.p
        L = p.max_read_length
        seqs = [r[2] for r in recs]
        packed, lens, W = H.pack_reads(seqs, L)
        lens_l = lens.tolist()
        n_pairs = len(recs) // 2
        # echo the reads as the reference prints them: decoded from the 2-bit codes (N -> G, truncated)
        out_recs = []
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        for r, rec in enumerate(recs):
            ln = lens_l[r]
            codes = H._CODE[np.frombuffer(rec[2], dtype=np.uint8)[:ln]]
            out_recs.append((rec[0], None if p.ignore_comments else rec[1], lut[codes].tobytes(), rec[3][:ln]))
        if self._first_batch:
            # SOAP4.cpp:465-474 / QueryParser.cpp:2253-2277: insert_low >= detected read lengths (first 10^6 samples)
            d1 = int(lens[0::2][:999999].max()) if n_pairs else 100
            d2 = int(lens[1::2][:999999].max()) if n_pairs else 100
            p.insert_low = max(p.insert_low, d1, d2)
            self._first_batch = False
        ctx = self.ctx
        ctx.set_reads(packed, lens)
        ap = H.AlignParams(p.insert_low, p.insert_high, p.clip_left, p.clip_right, p.mmp)
        hits, nh, rst = ctx.deep_dp_round(ap)
        self.stats.n_pairs += n_pairs
        self.stats.n_dp_instances += int(rst.n_dp_left) * 2
        w = _Writer(self.ix, p, out)
        done = np.zeros(n_pairs, dtype=bool)
        if nh:
            a = np.frombuffer(hits, dtype=np.dtype([("pair", "<u4"), ("lm", "<u4"), ("a1", "<u8"), ("a2", "<u8"), ("s1", "<i4"), ("s2", "<i4"),
                                                    ("st1", "<u4"), ("st2", "<u4"), ("d1", "<u2"), ("d2", "<u2"), ("c1", "<u2"), ("c2", "<u2")]), count=nh)
            # OutputBuffer::ready: sort by (pos1, pos2, score1, score2), drop duplicates (DV-DPfunctions.h:165-196)
            order = np.lexsort((a["s2"], a["s1"], a["a2"], a["a1"], a["pair"]))
            a = a[order]
            keep = np.ones(nh, dtype=bool)
            same = (a["pair"][1:] == a["pair"][:-1]) & (a["a1"][1:] == a["a1"][:-1]) & (a["a2"][1:] == a["a2"][:-1]) & \
                   (a["s1"][1:] == a["s1"][:-1]) & (a["s2"][1:] == a["s2"][:-1])
            keep[1:] = ~same
            a = a[keep]
            pid = a["pair"].astype(np.int64)
            c1 = _target_chr(self.ix, a["a1"], lens[2 * pid])
            c2 = _target_chr(self.ix, a["a2"], lens[2 * pid + 1])
            bounds = np.flatnonzero(np.concatenate([[True], pid[1:] != pid[:-1], [True]]))
            for b0, b1 in zip(bounds[:-1], bounds[1:]):
                q = int(pid[b0])
                res = [[int(a["a1"][i]), int(a["a2"][i]), int(a["s1"][i]), int(a["s2"][i])] for i in range(b0, b1)]
                w.pair(out_recs[2 * q], out_recs[2 * q + 1], res, c1[b0:b1].tolist(), c2[b0:b1].tolist())
                done[q] = True
        self.stats.n_deep_pairs += int(done.sum())
        failed = np.flatnonzero(~done)
        if len(failed):
            self._rescue(failed, out_recs, lens_l, w, ap)

    # ------------------------------------------------------------------ rescue passes for pairs the deep-DP round left
    def _rescue(self, failed, out_recs, lens, w, ap):
        """Single-end DP on the pooled seeds and mate rescue run on the device (`mp_rescue_round`); here only the
        per-pair bookkeeping of alignment.cpp:137-352: which pairs were rescued, which go out unpaired."""
        p = self.p
        (singles, ns), (resc, nr), rst = self.ctx.rescue_round(ap, p.skip_default_dp)
        self.stats.n_dp_instances += int(rst.n_single_tasks + rst.n_rescue_tasks)
        single = {}
        for i in range(ns):
            h = singles[i]
            single.setdefault(h.read_id, set()).add((h.algnmt, h.score))
        rescued = {}
        for i in range(nr):
            h = resc[i]
            if h.aligned_read & 1 == 0:
                rescued.setdefault(h.pair_id, set()).add((h.aligned_pos, h.dp_pos, h.aligned_score, h.dp_score))
            else:
                rescued.setdefault(h.pair_id, set()).add((h.dp_pos, h.aligned_pos, h.dp_score, h.aligned_score))
        # the default-DP stage writes every rescued pair of the batch before the unpaired output starts (alignment.cpp:225-330)
        for q in failed.tolist():
            if q not in rescued:
                continue
            r1, r2 = 2 * q, 2 * q + 1
            # OutputBuffer::ready(1): sorted by (pos1, pos2, score1, score2), duplicates and failed rescues dropped
            res = [list(x) for x in sorted(rescued[q])]
            c1 = _target_chr(self.ix, [x[0] for x in res], [lens[r1]] * len(res)).tolist()
            c2 = _target_chr(self.ix, [x[1] for x in res], [lens[r2]] * len(res)).tolist()
            w.pair(out_recs[r1], out_recs[r2], res, c1, c2)
            self.stats.n_rescued_pairs += 1
        for q in failed.tolist():
            if q in rescued:
                continue
            for r in (2 * q, 2 * q + 1):
                hs = sorted(single.get(r, ()))
                chrs = _target_chr(self.ix, [h[0] for h in hs], [lens[r]] * len(hs)).tolist()
                w.single(out_recs[r], [(c, h[1]) for c, h in zip(chrs, hs)])
            self.stats.n_unpaired_pairs += 1


def align_fastq(index: H.Index, params: Soap4Params, fq1: str, fq2: str, out, batch_pairs: int = 1 << 20) -> PathStats:
    """The body of `soap4 pair` (SOAP4.cpp:445-766): batches of <= 2 097 152 reads."""
    al = PairAligner(index, params)
    it1, it2 = read_fastq(fq1), read_fastq(fq2)
    batch = []
    while True:
        a, b = next(it1, None), next(it2, None)
        if (a is None) != (b is None):
            raise ValueError("Error: number of sequences of pair-end files not matched.")
        if a is None:
            break
        batch += [a, b]
        if len(batch) >= 2 * batch_pairs:
            al.align_batch(batch, out)
            batch = []
    if batch:
        al.align_batch(batch, out)
    st = al.stats
    al.close()
    return st
