"""ctypes bindings for the CPU oracle (oracle/libmp_oracle.so) and for the real-reference shims in oracle/_ref.

Test infrastructure only.  Nothing under megapath_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

u64 = C.c_uint64
u32 = C.c_uint32
u8p = C.POINTER(C.c_uint8)


class MmpParams(C.Structure):
    _fields_ = [("sa_size_threshold", C.c_int), ("seed_min_length", C.c_int), ("uniq_threshold", C.c_int),
                ("indel_fuzz", C.c_int), ("good_seed_len", C.c_int), ("reseed_len", C.c_int),
                ("reseed_rlt_ratio", C.c_double), ("reseed_abs_diff", C.c_int), ("short_seed_ratio", C.c_double)]


class SaSeed(C.Structure):
    _fields_ = [("query_offset", u32), ("sa_l", u64), ("sa_diff", u32), ("seed_len", u32)]


class SeedPos(C.Structure):
    _fields_ = [("pos", u64), ("strand_read_id", u32), ("paired_seed_length", u32)]


class SeedCounters(C.Structure):
    _fields_ = [("n_occ", u64), ("n_lkt", u64), ("n_sa", u64), ("n_seeds", u64), ("n_hits", u64)]


class DpResult(C.Structure):
    _fields_ = [("score", C.c_int), ("hit_loc", C.c_int), ("max_score_count", C.c_int), ("n_ops", C.c_int)]


class AlignParams(C.Structure):
    _fields_ = [("max_read_length", C.c_int), ("insert_low", C.c_int), ("insert_high", C.c_int),
                ("clip_left", C.c_int), ("clip_right", C.c_int), ("megapath_mode", C.c_int),
                ("top_percentage", C.c_double), ("quality_constant", C.c_int), ("skip_default_dp", C.c_int),
                ("mmp", MmpParams), ("ignore_comments", C.c_int)]


class AlignStats(C.Structure):
    _fields_ = [("n_pairs", u64), ("n_candidates", u64), ("n_deep_dp_pairs", u64), ("n_read_seedpos", u64),
                ("n_mate_seedpos", u64), ("n_dp_instances", u64), ("dp_cells", u64), ("n_single_dp_reads", u64),
                ("n_default_dp_pairs", u64), ("n_unpaired_pairs", u64), ("seed", SeedCounters)]


def build_oracle() -> str:
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libmp_oracle.so"])
    return os.path.join(ORACLE_DIR, "libmp_oracle.so")


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.mpo_index_open.restype = C.c_void_p
        lib.mpo_index_open.argtypes = [C.c_char_p]
        lib.mpo_index_close.argtypes = [C.c_void_p]
        for f in ("mpo_text_length", "mpo_inverse_sa0", "mpo_sa_interval"):
            getattr(lib, f).restype = u64
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.mpo_cum_freq.restype = u64
        lib.mpo_cum_freq.argtypes = [C.c_void_p, C.c_int]
        lib.mpo_num_seqs.argtypes = [C.c_void_p]
        lib.mpo_seq_name.restype = C.c_char_p
        lib.mpo_seq_name.argtypes = [C.c_void_p, C.c_int]
        lib.mpo_occ.restype = u64
        lib.mpo_occ.argtypes = [C.c_void_p, u64, C.c_uint]
        lib.mpo_psi_minus.restype = u64
        lib.mpo_psi_minus.argtypes = [C.c_void_p, u64]
        lib.mpo_sa_value.restype = u64
        lib.mpo_sa_value.argtypes = [C.c_void_p, u64]
        lib.mpo_lkt_range.argtypes = [C.c_void_p, u32, C.POINTER(u64), C.POINTER(u64)]
        lib.mpo_text_base.restype = C.c_uint
        lib.mpo_text_base.argtypes = [C.c_void_p, u64]
        lib.mpo_chr_and_pos.argtypes = [C.c_void_p, u64, C.POINTER(u64), C.POINTER(C.c_uint)]
        lib.mpo_lkt_table.restype = C.POINTER(u64)
        lib.mpo_lkt_table.argtypes = [C.c_void_p]
        lib.mpo_sgdp.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DpResult), u8p, C.c_int]
        lib.mpo_dp_cutoff.argtypes = [C.c_int]
        lib.mpo_cigar_from_ops.argtypes = [u8p, C.c_int, C.c_char_p, C.c_int]
        lib.mpo_mmp.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.POINTER(MmpParams), C.POINTER(SaSeed), C.c_int,
                                C.POINTER(SeedCounters)]
        lib.mpo_seed_read.argtypes = [C.c_void_p, u8p, C.c_int, u32, C.POINTER(MmpParams), C.POINTER(SeedPos),
                                      C.POINTER(C.c_int), C.POINTER(SeedPos), C.POINTER(C.c_int), C.c_int,
                                      C.POINTER(SeedCounters)]
        lib.mpo_default_params.argtypes = [C.POINTER(AlignParams)]
        lib.mpo_sort_longer_seed_perm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.mpo_align_fastq.argtypes = [C.c_void_p, C.POINTER(AlignParams), C.c_char_p, C.c_char_p, C.c_char_p,
                                        C.c_char_p, C.POINTER(AlignStats)]
        lib.mpo_align_fastq_bam.argtypes = [C.c_void_p, C.POINTER(AlignParams), C.c_char_p, C.c_char_p, C.c_char_p,
                                            C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(AlignStats)]
        _oracle = lib
    return _oracle


def align_fastq_bam(index_prefix, params, fq1, fq2, out_fastq, bam_text_prefix, print_mdnm=False, read_group=None):
    """Whole path with the BAM side as text; returns {"dpout": [...], "gout": [...], "unpair": [...], "header": str}."""
    lib = oracle()
    ix = lib.mpo_index_open(index_prefix.encode())
    st = AlignStats()
    rc = lib.mpo_align_fastq_bam(ix, C.byref(params), fq1.encode(), fq2.encode(), out_fastq.encode(), None,
                                 bam_text_prefix.encode(), int(print_mdnm), read_group.encode() if read_group else None, C.byref(st))
    lib.mpo_index_close(ix)
    assert rc == 0
    out = {k: open("%s.%s.txt" % (bam_text_prefix, k)).read().splitlines() for k in ("dpout", "gout", "unpair")}
    out["header"] = open(bam_text_prefix + ".header.txt").read()
    return out


def default_params(L=151, insert_high=750, mode=1, top=95, nc=False) -> AlignParams:
    p = AlignParams()
    oracle().mpo_default_params(C.byref(p))
    p.max_read_length = L
    p.insert_high = insert_high
    p.megapath_mode = mode
    p.top_percentage = top / 100.0
    p.ignore_comments = 1 if nc else 0
    return p


def as_u8p(a: np.ndarray):
    return a.ctypes.data_as(u8p)


def sgdp(ref: np.ndarray, read: np.ndarray, clip_l: int, clip_r: int, cutoff=None):
    """Run the oracle DP on one instance; returns (score, hit_loc, count, cigar)."""
    lib = oracle()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    if cutoff is None:
        cutoff = lib.mpo_dp_cutoff(len(read))
    res = DpResult()
    ops = np.zeros(len(ref) + len(read) + 8, dtype=np.uint8)
    lib.mpo_sgdp(as_u8p(ref), len(ref), as_u8p(read), len(read), clip_l, clip_r, cutoff, C.byref(res), as_u8p(ops), len(ops))
    cig = b""
    if res.score > 0:
        buf = C.create_string_buffer(4 * len(ops) + 16)
        lib.mpo_cigar_from_ops(as_u8p(ops), res.n_ops, buf, len(buf))
        cig = buf.value
    return res.score, res.hit_loc, res.max_score_count, cig.decode()


# ----------------------------------------------------------------------------- real reference (oracle/_ref)
def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libref_dp.so"))


_ref_dp = None
_ref_bwt = None


def ref_dp():
    global _ref_dp
    if _ref_dp is None:
        lib = C.CDLL(os.path.join(REF_DIR, "libref_dp.so"))
        lib.ref_callDP.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _ref_dp = lib
    return _ref_dp


def ref_bwt():
    global _ref_bwt
    if _ref_bwt is None:
        lib = C.CDLL(os.path.join(REF_DIR, "libref_bwt.so"))
        lib.ref_index_open.restype = C.c_void_p
        lib.ref_index_open.argtypes = [C.c_char_p, C.c_int]
        for f in ("ref_text_length", "ref_inverse_sa0", "ref_sa_interval"):
            getattr(lib, f).restype = u64
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.ref_occ_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_void_p]
        lib.ref_sa_many.argtypes = [C.c_void_p, C.c_void_p, u64, C.c_void_p]
        lib.ref_psi_minus.restype = u64
        lib.ref_psi_minus.argtypes = [C.c_void_p, u64]
        lib.ref_lkt_range.argtypes = [C.c_void_p, C.c_uint, C.POINTER(u64), C.POINTER(u64)]
        _ref_bwt = lib
    return _ref_bwt


def pack_dp_batch(seqs, max_len):
    """Pack sequences (arrays of codes) in the reference's DP input layout (DV-DPfunctions.cpp:3009-3072):
    32-instance interleave, symbol i (1-based) of instance t in word (t/32)*32*W + t%32 + 32*(i/16), bits 2*(15-i%16)."""
    n = len(seqs)
    W = (max_len + 15) >> 4
    out = np.zeros(((n + 31) // 32) * 32 * W + 64, dtype=np.uint32)
    for t, s in enumerate(seqs):
        base = (t // 32) * 32 * W + (t % 32)
        for i, c in enumerate(s, start=1):
            out[base + 32 * (i >> 4)] |= np.uint32(int(c) << (2 * (15 - (i & 15))))
    return out


def decode_pattern(pat: bytes) -> str:
    """Reference pattern bytes (CPU_DP.cpp:622-786, decoded as at DV-DPfunctions.cpp:3449-3461) -> special CIGAR."""
    runs = []
    last = None
    i = 0
    while i < len(pat) and pat[i] != 0:
        ch = pat[i]
        if ch == ord("V"):
            i += 1
            cnt = pat[i] - 1
            t = last
        else:
            t = chr(ch)
            cnt = 1
            last = t
        if runs and runs[-1][0] == t:
            runs[-1][1] += cnt
        else:
            runs.append([t, cnt])
        i += 1
    return "".join("%d%s" % (c, t) for t, c in reversed(runs) if c > 0)


def ref_call_dp(refs, reads, clip_l, clip_r, cutoffs=None):
    """Run the real callDP on lists of code arrays; returns list of (score, hit_loc, count, cigar)."""
    lib = ref_dp()
    n = len(refs)
    max_dna = max(len(r) for r in refs) + 1
    max_read = max(len(r) for r in reads) + 1
    max_dna = (max_dna + 3) // 4 * 4
    max_read = (max_read + 3) // 4 * 4
    dna = pack_dp_batch(refs, max_dna)
    rd = pack_dp_batch(reads, max_read)
    dl = np.array([len(r) for r in refs], dtype=np.uint32)
    rl = np.array([len(r) for r in reads], dtype=np.uint32)
    if cutoffs is None:
        cutoffs = [oracle().mpo_dp_cutoff(int(x)) for x in rl]
    co = np.array(cutoffs, dtype=np.int32)
    sc = np.zeros(n, dtype=np.int32)
    hl = np.zeros(n, dtype=np.uint32)
    mc = np.zeros(n, dtype=np.uint32)
    stride = max_dna + max_read
    pat = np.zeros(n * stride, dtype=np.uint8)
    rc = lib.ref_callDP(dna.ctypes.data, dl.ctypes.data, max_dna, rd.ctypes.data, rl.ctypes.data, max_read, clip_l, clip_r,
                        -2, -3, co.ctypes.data, n, sc.ctypes.data, hl.ctypes.data, mc.ctypes.data, pat.ctypes.data)
    assert rc == 0
    out = []
    for t in range(n):
        cig = decode_pattern(pat[t * stride:(t + 1) * stride].tobytes()) if sc[t] > 0 else ""
        out.append((int(sc[t]), int(hl[t]), int(mc[t]), cig))
    return out
