"""Helpers shared by the -m gpu parity tests: read FASTQ fixtures, drive the HIP library through its C ABI
(megapath_amd.hipapi) and the oracle (oracle_lib) on the same inputs."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O
from megapath_amd import hipapi as H

_CODE = H._CODE


def read_fastq_pairs(fq1, fq2):
    def rd(p):
        with open(p, "rb") as f:
            lines = f.read().split(b"\n")
        return [lines[i + 1] for i in range(0, len(lines) - 3, 4)]
    a, b = rd(fq1), rd(fq2)
    seqs = []
    for x, y in zip(a, b):
        seqs += [x, y]
    return seqs


def codes_of(seq: bytes, L: int) -> np.ndarray:
    return _CODE[np.frombuffer(seq, dtype=np.uint8)[: L - 1]].astype(np.uint8)


def oracle_seeds(ix, seqs, L, mmp=None):
    """Per-read kept SeedPos from the oracle: list of [(pos, strand_read_id, covered)] ('+' first, ascending pos)."""
    lib = O.oracle()
    p = O.default_params(L=L)
    if mmp is not None:
        p.mmp = mmp
    out = []
    ctr = O.SeedCounters()
    cap = 400 * 64
    sp = (O.SeedPos * cap)()
    sn = (O.SeedPos * cap)()
    for r, s in enumerate(seqs):
        codes = np.ascontiguousarray(codes_of(s, L))
        npos, nneg = C.c_int(), C.c_int()
        lib.mpo_seed_read(ix, O.as_u8p(codes), len(codes), r & ~1, C.byref(p.mmp), sp, C.byref(npos), sn, C.byref(nneg), cap, C.byref(ctr))
        rec = [(sp[i].pos, sp[i].strand_read_id, sp[i].paired_seed_length) for i in range(npos.value)]
        rec += [(sn[i].pos, sn[i].strand_read_id, sn[i].paired_seed_length) for i in range(nneg.value)]
        out.append(rec)
    return out, ctr


def to_oracle_mmp(m: H.MmpParams) -> O.MmpParams:
    return O.MmpParams(m.sa_size_threshold, m.seed_min_length, m.uniq_threshold, m.indel_fuzz, m.good_seed_len, m.reseed_len,
                       m.reseed_rlt_ratio, m.reseed_abs_diff, m.short_seed_ratio)
