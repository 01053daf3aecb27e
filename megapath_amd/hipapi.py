"""ctypes mirror of include/megapath_hip.h.  No CPU fallback: if the library or a gfx950 device is missing, every
entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmegapath_hip.so")


class HipLibraryMissing(RuntimeError):
    pass


class MpError(RuntimeError):
    pass


class MmpParams(C.Structure):
    _fields_ = [("sa_size_threshold", C.c_int32), ("seed_min_length", C.c_int32), ("uniq_threshold", C.c_int32),
                ("indel_fuzz", C.c_int32), ("good_seed_len", C.c_int32), ("reseed_len", C.c_int32),
                ("reseed_rlt_ratio", C.c_double), ("reseed_abs_diff", C.c_int32), ("short_seed_ratio", C.c_double)]

    @classmethod
    def nt2(cls):
        """[MMP] section of soap4-nt2.ini (reference soap4/soap4-nt2.ini:114-123)."""
        return cls(30, 17, 6, 5, 27, 18, 0.7, 4, 0.5)

    @classmethod
    def human(cls):
        """[MMP] section of soap4.ini (seed min length 22, reseed length 23)."""
        return cls(30, 22, 6, 5, 27, 23, 0.7, 4, 0.5)


class SeedPos(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("strand_read_id", C.c_uint32), ("paired_seed_length", C.c_uint32)]


class SeedStats(C.Structure):
    _fields_ = [("n_occ", C.c_uint64), ("n_lkt", C.c_uint64), ("n_sa", C.c_uint64), ("n_seeds", C.c_uint64),
                ("n_hits", C.c_uint64), ("n_seedpos", C.c_uint64), ("ms_seed", C.c_float), ("ms_locate", C.c_float),
                ("ms_merge", C.c_float), ("reserved", C.c_float), ("n_prefilter", C.c_uint64), ("n_occ_loaded", C.c_uint64),
                ("n_lkt16", C.c_uint64), ("n_prefilter_loads", C.c_uint64), ("n_text_words", C.c_uint64), ("n_steps_skipped", C.c_uint64), ("n_sa_seed_kernel", C.c_uint64), ("n_lf_steps", C.c_uint64)]


class DpCensus(C.Structure):
    _fields_ = [("instances", C.c_uint64), ("cells_full_table", C.c_uint64), ("cells_band_score", C.c_uint64),
                ("cells_band_trace", C.c_uint64), ("cells_full_score", C.c_uint64), ("cells_full_trace", C.c_uint64)]


class DpTask(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("strand", C.c_uint32), ("win_start", C.c_uint64), ("win_len", C.c_uint32),
                ("reserved", C.c_uint32)]


DP_TASK_DTYPE = np.dtype([("read_id", "<u4"), ("strand", "<u4"), ("win_start", "<u8"), ("win_len", "<u4"), ("reserved", "<u4")])


class DpResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("hit_loc", C.c_uint32), ("max_score_count", C.c_uint32), ("n_ops", C.c_uint32)]


class AlignParams(C.Structure):
    _fields_ = [("insert_low", C.c_int32), ("insert_high", C.c_int32), ("clip_left", C.c_int32), ("clip_right", C.c_int32),
                ("mmp", MmpParams), ("want_tie_counts", C.c_int32), ("reserved", C.c_int32)]


class PairHit(C.Structure):
    _fields_ = [("pair_id", C.c_uint32), ("left_is_mate", C.c_uint32), ("algnmt_1", C.c_uint64), ("algnmt_2", C.c_uint64),
                ("score_1", C.c_int32), ("score_2", C.c_int32), ("start_1", C.c_uint32), ("start_2", C.c_uint32),
                ("dplen_1", C.c_uint16), ("dplen_2", C.c_uint16), ("count_1", C.c_uint16), ("count_2", C.c_uint16)]


class RoundStats(C.Structure):
    _fields_ = [("seed", SeedStats), ("n_candidates", C.c_uint64), ("n_dp_left", C.c_uint64), ("n_dp_right", C.c_uint64),
                ("n_pair_hits", C.c_uint64), ("dp_cells", C.c_uint64), ("ms_pair", C.c_float), ("ms_dp_fill", C.c_float),
                ("ms_dp_trace", C.c_float), ("ms_total", C.c_float)]


class SingleHit(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("strand", C.c_uint32), ("algnmt", C.c_uint64), ("score", C.c_int32), ("count", C.c_uint32),
                ("win_start", C.c_uint64), ("win_len", C.c_uint32), ("reserved", C.c_uint32)]


class RescueHit(C.Structure):
    _fields_ = [("pair_id", C.c_uint32), ("aligned_read", C.c_uint32), ("aligned_pos", C.c_uint64), ("dp_pos", C.c_uint64),
                ("aligned_score", C.c_int32), ("dp_score", C.c_int32), ("aligned_strand", C.c_uint32), ("dp_strand", C.c_uint32),
                ("aligned_win_start", C.c_uint64), ("dp_win_start", C.c_uint64), ("aligned_win_len", C.c_uint32), ("dp_win_len", C.c_uint32),
                ("aligned_count", C.c_uint32), ("dp_count", C.c_uint32)]


class RescueStats(C.Structure):
    _fields_ = [("n_failed_pairs", C.c_uint64), ("n_single_tasks", C.c_uint64), ("n_single_hits", C.c_uint64), ("n_rescue_tasks", C.c_uint64),
                ("n_rescue_hits", C.c_uint64), ("ms_tasks", C.c_float), ("ms_dp_fill", C.c_float), ("ms_dp_trace", C.c_float), ("ms_total", C.c_float),
                ("n_floor_pairs", C.c_uint64), ("n_floor_tasks", C.c_uint64), ("n_floor_mate_tried", C.c_uint64), ("n_floor_mate_pairs", C.c_uint64)]


class IndexInfo(C.Structure):
    _fields_ = [("text_length", C.c_uint64), ("inverse_sa0", C.c_uint64), ("cum_freq", C.c_uint64 * 5),
                ("sa_interval", C.c_uint32), ("num_seq", C.c_uint32), ("device_bytes", C.c_uint64), ("prefilter_k", C.c_uint32),
                ("reserved", C.c_uint32)]


EXPORTS = ["mp_index_open", "mp_index_from_arrays", "mp_index_clone", "mp_index_close", "mp_index_get_info", "mp_index_seq_name",
           "mp_index_chr_and_pos", "mp_index_chr_and_pos_batch", "mp_index_chr_pos_segment", "mp_index_target_chr", "mp_index_seq_info", "mp_index_fetch_text", "mp_index_occ", "mp_index_sa", "mp_index_lkt", "mp_index_copy_lkt", "mp_index_prefilter", "mp_index_prefilter3", "mp_index_check_lkt16", "mp_ctx_create", "mp_ctx_destroy", "mp_ctx_set_index",
           "mp_set_reads", "mp_set_reads_device", "mp_seed_batch", "mp_ctx_set_prefilter", "mp_ctx_set_dp_band", "mp_test_dp_band", "mp_ctx_dp_band_stats", "mp_ctx_set_dp_thermo", "mp_ctx_dp_thermo_stats", "mp_test_dp_thermo", "mp_test_dp_thermo_kind", "mp_ctx_set_dp_census", "mp_ctx_get_dp_census", "mp_get_seeds", "mp_sgdp_batch", "mp_deep_dp_round",
           "mp_get_pair_hits", "mp_get_pair_hits_view", "mp_ctx_set_hit_copies", "mp_reduce_begin", "mp_reduce_chunk", "mp_reduce_chunk_part", "mp_ctx_set_score_floor", "mp_chunk_stash", "mp_test_floor_bound", "mp_test_floor_mate_bound", "mp_chunk_unstash", "mp_reduce_finish", "mp_rescue_round", "mp_get_single_hits", "mp_get_rescue_hits", "mp_get_single_hits_view", "mp_get_rescue_hits_view", "mp_bench_gather64", "mp_device_check", "mp_device_memory", "mp_strerror",
           "mp_test_sort_longer_seed", "mp_coverage_create", "mp_coverage_destroy", "mp_coverage_add", "mp_coverage_finish", "mp_coverage_seq_stats",
           "mp_coverage_spike_intervals", "mp_coverage_spike_bases", "mp_build_index_arrays"]


def build_index_arrays_native(codes, sa_interval: int = 16, device: int = 0):
    """The compute of the reference's 2bwt-builder on the device (mp_build_index_arrays): from the text's base codes the BWT words
    (MSB-first, '$' row left out), inverseSa0, cumulative counts, sampled suffix array and .pac bytes -- host numpy arrays in the
    layouts megapath_amd.index_builder.write_index_files writes.  Returns (dict, info) with info = {rounds, ms_suffix_sort}."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n = int(codes.size)
    bwt_words = np.zeros((n + 15) // 16, dtype=np.uint32)
    samples = np.zeros((n + 1 + sa_interval - 1) // sa_interval, dtype=np.uint64)
    pac = np.zeros((n + 3) // 4, dtype=np.uint8)
    inv, cum = C.c_uint64(), (C.c_uint64 * 5)()
    rounds, ms = C.c_uint32(), C.c_float()
    lib = load_library()
    lib.mp_build_index_arrays.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    _chk(lib.mp_build_index_arrays(device, codes.ctypes.data, n, sa_interval, bwt_words.ctypes.data, samples.ctypes.data, pac.ctypes.data, C.byref(inv), cum,
                                   C.byref(rounds), C.byref(ms)), "mp_build_index_arrays")
    return (dict(n=n, inv_sa0=int(inv.value), cum=np.array(list(cum), dtype=np.uint64), bwt_words=bwt_words, sa_samples=samples, pac=pac),
            dict(rounds=int(rounds.value), ms_suffix_sort=float(ms.value)))


def build_library(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", src])
    return LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    lib.mp_index_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.mp_index_from_arrays.argtypes = [C.c_int, C.c_uint64, C.c_uint64, u64p, vp, vp, C.c_uint32, vp, vp, C.c_uint32,
                                         C.POINTER(C.c_char_p), vp, vp, C.POINTER(vp)]
    lib.mp_index_close.argtypes = [vp]
    lib.mp_index_clone.argtypes = [vp, C.POINTER(vp)]
    lib.mp_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    lib.mp_index_seq_name.restype = C.c_char_p
    lib.mp_index_seq_name.argtypes = [vp, C.c_uint32]
    lib.mp_index_chr_and_pos.argtypes = [vp, C.c_uint64, u64p, u32p]
    lib.mp_index_chr_and_pos_batch.argtypes = [vp, vp, C.c_uint64, vp, vp]
    lib.mp_index_occ.argtypes = [vp, vp, vp, C.c_uint64, vp]
    lib.mp_index_sa.argtypes = [vp, vp, C.c_uint64, vp]
    lib.mp_index_lkt.argtypes = [vp, vp, C.c_uint64, vp, vp]
    lib.mp_index_prefilter.argtypes = [vp, vp, C.c_uint64, vp]
    lib.mp_index_copy_lkt.argtypes = [vp, vp]
    lib.mp_index_chr_pos_segment.argtypes = [vp, C.c_uint64, u64p, u32p, u64p]
    lib.mp_index_target_chr.argtypes = [vp, C.c_uint64, C.c_uint32, u32p]
    lib.mp_index_seq_info.argtypes = [vp, C.c_uint32, u64p, u64p, u64p]
    lib.mp_index_fetch_text.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.mp_index_check_lkt16.argtypes = [vp, C.c_uint64, u64p, u64p]
    lib.mp_index_prefilter3.argtypes = [vp, vp, C.c_uint64, vp]
    lib.mp_ctx_set_prefilter.argtypes = [vp, C.c_int]
    lib.mp_ctx_set_dp_band.argtypes = [vp, C.c_int]
    lib.mp_test_dp_band.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp]
    lib.mp_ctx_dp_band_stats.argtypes = [vp, u64p, u64p]
    lib.mp_ctx_set_dp_census.argtypes = [vp, C.c_int]
    lib.mp_ctx_set_dp_thermo.argtypes = [vp, C.c_int]
    lib.mp_ctx_dp_thermo_stats.argtypes = [vp, vp]
    lib.mp_test_dp_thermo.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.mp_test_dp_thermo_kind.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, vp]
    lib.mp_ctx_get_dp_census.argtypes = [vp, C.POINTER(DpCensus)]
    lib.mp_ctx_create.argtypes = [vp, C.POINTER(vp)]
    lib.mp_ctx_destroy.argtypes = [vp]
    lib.mp_ctx_set_index.argtypes = [vp, vp]
    lib.mp_set_reads.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.mp_set_reads_device.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    lib.mp_seed_batch.argtypes = [vp, C.POINTER(MmpParams), C.POINTER(SeedStats)]
    lib.mp_get_seeds.argtypes = [vp, vp, C.c_uint64, vp]
    lib.mp_sgdp_batch.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mp_deep_dp_round.argtypes = [vp, C.POINTER(AlignParams), u64p, C.POINTER(RoundStats)]
    lib.mp_get_pair_hits.argtypes = [vp, vp, C.c_uint64]
    lib.mp_get_pair_hits_view.argtypes = [vp, C.POINTER(vp), u64p]
    lib.mp_rescue_round.argtypes = [vp, C.POINTER(AlignParams), C.c_int, C.POINTER(RescueStats)]
    lib.mp_get_single_hits.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.mp_get_rescue_hits.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.mp_get_single_hits_view.argtypes = [vp, C.POINTER(vp), u64p]
    lib.mp_get_rescue_hits_view.argtypes = [vp, C.POINTER(vp), u64p]
    lib.mp_ctx_set_hit_copies.argtypes = [vp, C.c_int]
    lib.mp_reduce_begin.argtypes = [vp, C.c_uint32]
    lib.mp_reduce_chunk.argtypes = [vp, C.c_int, C.c_uint32, vp, vp]
    lib.mp_reduce_chunk_part.argtypes = [vp, C.c_int, C.c_uint32, C.c_int, vp, vp]
    lib.mp_ctx_set_score_floor.argtypes = [vp, C.c_int]
    lib.mp_chunk_stash.argtypes = [vp, C.c_uint32]
    lib.mp_test_floor_bound.argtypes = [vp, vp, C.c_uint32, vp]
    lib.mp_test_floor_mate_bound.argtypes = [vp, vp, C.c_uint32, C.c_int, vp]
    lib.mp_chunk_unstash.argtypes = [vp, C.c_uint32]
    lib.mp_reduce_finish.argtypes = [vp, C.POINTER(vp), u64p]
    lib.mp_bench_gather64.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), u64p]
    lib.mp_device_check.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    lib.mp_test_sort_longer_seed.argtypes = [C.c_int, vp, vp, C.c_uint32, vp]
    lib.mp_coverage_create.argtypes = [vp, C.POINTER(vp)]
    lib.mp_coverage_destroy.argtypes = [vp]
    lib.mp_coverage_add.argtypes = [vp, vp, vp, vp, C.c_uint64]
    lib.mp_coverage_finish.argtypes = [vp, C.c_double]
    lib.mp_coverage_seq_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mp_coverage_spike_intervals.argtypes = [vp, vp, vp, vp, C.c_uint64, u64p]
    lib.mp_coverage_spike_bases.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
    lib.mp_strerror.restype = C.c_char_p
    lib.mp_strerror.argtypes = [C.c_int]
    _lib = lib
    return lib


def _chk(rc: int, what: str):
    if rc != 0:
        raise MpError("%s failed: %s (%d)" % (what, load_library().mp_strerror(rc).decode(), rc))


_CODE = np.zeros(256, dtype=np.uint8)           # INDEXFillCharMap (IndexHandler.cpp:26-46): others -> A, N -> G
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i
    _CODE[_ch + 32] = _i
_CODE[ord("U")] = _CODE[ord("u")] = 3
_CODE[ord("N")] = _CODE[ord("n")] = 2


def sort_longer_seed(seed_len_lists, device: int = 0):
    """Test aid: the rank order single_cand_kernel gives each list of seed lengths (indices into the list)."""
    off = np.zeros(len(seed_len_lists) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in seed_len_lists])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in seed_len_lists]) if len(seed_len_lists) else np.zeros(0, np.uint32))
    perm = np.zeros(max(int(off[-1]), 1), dtype=np.uint32)
    _chk(load_library().mp_test_sort_longer_seed(device, flat.ctypes.data, off.ctypes.data, len(seed_len_lists), perm.ctypes.data), "mp_test_sort_longer_seed")
    return [perm[off[i]:off[i + 1]].copy() for i in range(len(seed_len_lists))]


def pack_reads(seqs, max_read_length: int):
    """Host-side read packing as appendToQueryArrays (QueryParser.cpp:184-234): truncate to max_read_length-1, map
    characters, 2 bits per base LSB-first; each read's words contiguous.  Returns (packed[u32 n x W], lengths, W)."""
    n = len(seqs)
    W = (max_read_length + 15) // 16
    packed = np.zeros((n, W), dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint32)
    for r, s in enumerate(seqs):
        a = np.frombuffer(s, dtype=np.uint8)[: max_read_length - 1]
        codes = _CODE[a].astype(np.uint32)
        lens[r] = len(codes)
        pad = np.zeros(W * 16, dtype=np.uint32)
        pad[: len(codes)] = codes
        packed[r] = (pad.reshape(W, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    return packed, lens, W


class Index:
    def __init__(self, prefix: str, device: int = 0):
        lib = load_library()
        self._h = C.c_void_p()
        _chk(lib.mp_index_open(prefix.encode(), device, C.byref(self._h)), "mp_index_open")
        self.info = IndexInfo()
        _chk(lib.mp_index_get_info(self._h, C.byref(self.info)), "mp_index_get_info")

    @classmethod
    def from_arrays(cls, device, n, inv_sa0, cum, bwt_words, sa_samples, sa_interval, pac, names, starts, lengths):
        lib = load_library()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        cumv = (C.c_uint64 * 5)(*[int(x) for x in cum])
        bwt_words = np.ascontiguousarray(bwt_words, dtype=np.uint32)
        sa_samples = np.ascontiguousarray(sa_samples, dtype=np.uint64)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        cn = (C.c_char_p * len(names))(*[x.encode() for x in names])
        _chk(lib.mp_index_from_arrays(device, int(n), int(inv_sa0), cumv, bwt_words.ctypes.data, sa_samples.ctypes.data,
                                      int(sa_interval), pac.ctypes.data, None, len(names), cn, starts.ctypes.data,
                                      lengths.ctypes.data, C.byref(self._h)), "mp_index_from_arrays")
        self.info = IndexInfo()
        _chk(lib.mp_index_get_info(self._h, C.byref(self.info)), "mp_index_get_info")
        return self

    def clone(self):
        """A second copy of this index in the same GPU's HBM (mp_index_clone: device-to-device copies)."""
        lib = load_library()
        other = type(self).__new__(type(self))
        other._h = C.c_void_p()
        _chk(lib.mp_index_clone(self._h, C.byref(other._h)), "mp_index_clone")
        other.info = IndexInfo()
        _chk(lib.mp_index_get_info(other._h, C.byref(other.info)), "mp_index_get_info")
        return other

    def close(self):
        if self._h:
            load_library().mp_index_close(self._h)
            self._h = None

    def seq_name(self, seq_id: int) -> str:
        return load_library().mp_index_seq_name(self._h, seq_id).decode()

    def chr_and_pos(self, amb_pos: int):
        p, c = C.c_uint64(), C.c_uint32()
        _chk(load_library().mp_index_chr_and_pos(self._h, int(amb_pos), C.byref(p), C.byref(c)), "mp_index_chr_and_pos")
        return p.value, c.value

    def chr_and_pos_batch(self, amb_pos):
        amb_pos = np.ascontiguousarray(amb_pos, dtype=np.uint64)
        p = np.zeros(len(amb_pos), dtype=np.uint64)
        c = np.zeros(len(amb_pos), dtype=np.uint32)
        _chk(load_library().mp_index_chr_and_pos_batch(self._h, amb_pos.ctypes.data, len(amb_pos), p.ctypes.data, c.ctypes.data),
             "mp_index_chr_and_pos_batch")
        return p, c

    def occ(self, idx, sym):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        sym = np.ascontiguousarray(sym, dtype=np.uint32)
        out = np.zeros(len(idx), dtype=np.uint64)
        _chk(load_library().mp_index_occ(self._h, idx.ctypes.data, sym.ctypes.data, len(idx), out.ctypes.data), "mp_index_occ")
        return out

    def sa(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros(len(idx), dtype=np.uint64)
        _chk(load_library().mp_index_sa(self._h, idx.ctypes.data, len(idx), out.ctypes.data), "mp_index_sa")
        return out

    def bench_gather64(self, lanes=1 << 21, per_lane=64, repeats=5):
        ms, nb = C.c_float(), C.c_uint64()
        _chk(load_library().mp_bench_gather64(self._h, lanes, per_lane, repeats, C.byref(ms), C.byref(nb)), "mp_bench_gather64")
        return ms.value, nb.value

    def lkt_table(self):
        """The resident 13-mer table: 4^13 inclusive right SA bounds (uint32)."""
        out = np.zeros(4 ** 13, dtype=np.uint32)
        _chk(load_library().mp_index_copy_lkt(self._h, out.ctypes.data), "mp_index_copy_lkt")
        return out

    def prefilter(self, keys):
        """Bits of the seeding kernel's 17-mer pre-filter for 34-bit keys (first scanned base in the low bits)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint8)
        _chk(load_library().mp_index_prefilter(self._h, keys.ctypes.data, len(keys), out.ctypes.data), "mp_index_prefilter")
        return out

    def prefilter3(self, keys):
        """Per key the answers of the three alignments of the per-15-mer pre-filter table (bits 0..2)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint8)
        _chk(load_library().mp_index_prefilter3(self._h, keys.ctypes.data, len(keys), out.ctypes.data), "mp_index_prefilter3")
        return out

    def check_lkt16(self, n_random: int = 0):
        """(keys checked, mismatches) of the 16-mer look-up table against 13-mer table + three backward steps."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _chk(load_library().mp_index_check_lkt16(self._h, n_random, C.byref(a), C.byref(b)), "mp_index_check_lkt16")
        return int(a.value), int(b.value)

    def lkt(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        l = np.zeros(len(keys), dtype=np.uint64)
        r = np.zeros(len(keys), dtype=np.uint64)
        _chk(load_library().mp_index_lkt(self._h, keys.ctypes.data, len(keys), l.ctypes.data, r.ctypes.data), "mp_index_lkt")
        return l, r


class Coverage:
    """Per-reference coverage and spike filter on the device (mp_coverage_*)."""

    def __init__(self, index: Index):
        self._h = C.c_void_p()
        _chk(load_library().mp_coverage_create(index._h, C.byref(self._h)), "mp_coverage_create")

    @staticmethod
    def _arrs(seq, start, span):
        return (np.ascontiguousarray(seq, dtype=np.uint32), np.ascontiguousarray(start, dtype=np.uint64), np.ascontiguousarray(span, dtype=np.uint32))

    def add(self, seq_id_1based, start0, span):
        a, b, c = self._arrs(seq_id_1based, start0, span)
        _chk(load_library().mp_coverage_add(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a)), "mp_coverage_add")

    def finish(self, stdev_k: float):
        _chk(load_library().mp_coverage_finish(self._h, float(stdev_k)), "mp_coverage_finish")

    def seq_stats(self, seq_id_1based: int):
        m, v, x = C.c_double(), C.c_double(), C.c_double()
        _chk(load_library().mp_coverage_seq_stats(self._h, seq_id_1based, C.byref(m), C.byref(v), C.byref(x)), "mp_coverage_seq_stats")
        return m.value, v.value, x.value

    def spike_intervals(self):
        n = C.c_uint64()
        _chk(load_library().mp_coverage_spike_intervals(self._h, None, None, None, 0, C.byref(n)), "mp_coverage_spike_intervals")
        s = np.zeros(n.value + 1, dtype=np.uint32); a = np.zeros(n.value + 1, dtype=np.uint64); b = np.zeros(n.value + 1, dtype=np.uint64)
        _chk(load_library().mp_coverage_spike_intervals(self._h, s.ctypes.data, a.ctypes.data, b.ctypes.data, n.value, C.byref(n)), "mp_coverage_spike_intervals")
        return [(int(s[i]), int(a[i]), int(b[i])) for i in range(n.value)]

    def spike_bases(self, seq_id_1based, start0, span):
        a, b, c = self._arrs(seq_id_1based, start0, span)
        out = np.zeros(max(len(a), 1), dtype=np.uint32)
        _chk(load_library().mp_coverage_spike_bases(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, len(a), out.ctypes.data), "mp_coverage_spike_bases")
        return out[:len(a)]

    def close(self):
        if self._h:
            load_library().mp_coverage_destroy(self._h)
            self._h = None


class Context:
    def __init__(self, index: Index):
        self.index = index
        self._h = C.c_void_p()
        _chk(load_library().mp_ctx_create(index._h, C.byref(self._h)), "mp_ctx_create")
        self.n_reads = 0
        self._keep = None

    def close(self):
        if self._h:
            load_library().mp_ctx_destroy(self._h)
            self._h = None

    def set_index(self, index: Index):
        """Bind the context to another index chunk resident on the same GPU; the reads of the batch stay set."""
        _chk(load_library().mp_ctx_set_index(self._h, index._h), "mp_ctx_set_index")
        self.index = index

    def set_reads(self, packed: np.ndarray, lens: np.ndarray):
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.n_reads = len(lens)
        _chk(load_library().mp_set_reads(self._h, packed.ctypes.data, packed.shape[1], lens.ctypes.data, len(lens)), "mp_set_reads")

    def set_reads_device(self, d_packed_ptr: int, W: int, d_lens_ptr: int, n_reads: int, max_len: int, keep=None):
        self.n_reads = n_reads
        self._keep = keep
        _chk(load_library().mp_set_reads_device(self._h, d_packed_ptr, W, d_lens_ptr, n_reads, max_len), "mp_set_reads_device")

    def seed_batch(self, params: MmpParams) -> SeedStats:
        st = SeedStats()
        _chk(load_library().mp_seed_batch(self._h, C.byref(params), C.byref(st)), "mp_seed_batch")
        return st

    def set_prefilter(self, on: bool):
        _chk(load_library().mp_ctx_set_prefilter(self._h, int(bool(on))), "mp_ctx_set_prefilter")

    def set_dp_band(self, on: bool):
        _chk(load_library().mp_ctx_set_dp_band(self._h, int(bool(on))), "mp_ctx_set_dp_band")

    def set_dp_thermo(self, on):
        """0 / False off, 1 / True every shape the certificate has a variant for, 2 round 3's coverage (single-end windows of 101 .. 320-base reads)"""
        _chk(load_library().mp_ctx_set_dp_thermo(self._h, int(on)), "mp_ctx_set_dp_thermo")

    def dp_thermo_stats(self):
        """instances seen by the certificate kernel: (certified, overflow, bound too high, not eligible)"""
        out = (C.c_uint64 * 4)()
        _chk(load_library().mp_ctx_dp_thermo_stats(self._h, out), "mp_ctx_dp_thermo_stats")
        return tuple(int(x) for x in out)

    def test_dp_thermo(self, tasks_np: np.ndarray, kind: int = -1):
        """test hook: the certificate kernel on a numpy array of DP_TASK_DTYPE records -> (cert u8[n], stats); kind: the variant
        (mp_test_dp_thermo_kind: 0 short reads, 1 narrow, 2 wide, 3 / 4 mate-rescue windows; -1 by the longest read)"""
        assert tasks_np.dtype == DP_TASK_DTYPE and tasks_np.flags.c_contiguous
        n = len(tasks_np)
        cert = np.zeros(max(n, 1), dtype=np.uint8)
        st = (C.c_uint64 * 4)()
        _chk(load_library().mp_test_dp_thermo_kind(self._h, tasks_np.ctypes.data, n, kind, cert.ctypes.data, st), "mp_test_dp_thermo_kind")
        return cert[:n], tuple(int(x) for x in st)

    def sgdp_scores(self, tasks_np: np.ndarray, clip_left: int, clip_right: int) -> np.ndarray:
        """exact DP scores (no operations) of a numpy array of DP_TASK_DTYPE records"""
        assert tasks_np.dtype == DP_TASK_DTYPE and tasks_np.flags.c_contiguous
        n = len(tasks_np)
        res = np.zeros(max(n, 1), dtype=np.dtype([("score", "<i4"), ("hit_loc", "<u4"), ("max_score_count", "<u4"), ("n_ops", "<u4")]))
        f, t = C.c_float(), C.c_float()
        _chk(load_library().mp_sgdp_batch(self._h, tasks_np.ctypes.data, n, clip_left, clip_right, res.ctypes.data, None, 0, C.byref(f), C.byref(t)), "mp_sgdp_batch")
        return res["score"][:n].copy()

    def sgdp_results(self, tasks_np: np.ndarray, clip_left: int, clip_right: int, band: bool = False) -> np.ndarray:
        """(score, hit_loc, max_score_count) of a numpy array of DP_TASK_DTYPE records: band=False the full-table kernels (mp_sgdp_batch), band=True
        the band scheme of the deep-DP round with its pre-filter short cuts (test hook mp_test_dp_band)"""
        assert tasks_np.dtype == DP_TASK_DTYPE and tasks_np.flags.c_contiguous
        n = len(tasks_np)
        res = np.zeros(max(n, 1), dtype=np.dtype([("score", "<i4"), ("hit_loc", "<u4"), ("max_score_count", "<u4"), ("n_ops", "<u4")]))
        lib = load_library()
        if band:
            _chk(lib.mp_test_dp_band(self._h, tasks_np.ctypes.data, n, clip_left, clip_right, res.ctypes.data), "mp_test_dp_band")
        else:
            f, t = C.c_float(), C.c_float()
            _chk(lib.mp_sgdp_batch(self._h, tasks_np.ctypes.data, n, clip_left, clip_right, res.ctypes.data, None, 0, C.byref(f), C.byref(t)), "mp_sgdp_batch")
        return res[:n]

    def set_dp_census(self, on: bool):
        _chk(load_library().mp_ctx_set_dp_census(self._h, int(bool(on))), "mp_ctx_set_dp_census")

    def dp_census(self) -> DpCensus:
        out = DpCensus()
        _chk(load_library().mp_ctx_get_dp_census(self._h, C.byref(out)), "mp_ctx_get_dp_census")
        return out

    def dp_band_stats(self):
        """(instances run through the band kernel, instances of those re-run by the full kernels)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _chk(load_library().mp_ctx_dp_band_stats(self._h, C.byref(a), C.byref(b)), "mp_ctx_dp_band_stats")
        return int(a.value), int(b.value)

    def get_seeds(self):
        lib = load_library()
        off = np.zeros(self.n_reads + 1, dtype=np.uint64)
        _chk(lib.mp_get_seeds(self._h, None, 0, off.ctypes.data), "mp_get_seeds")
        n = int(off[-1])
        arr = (SeedPos * max(n, 1))()
        _chk(lib.mp_get_seeds(self._h, arr, n, off.ctypes.data), "mp_get_seeds")
        return arr, off

    def sgdp_batch(self, tasks, clip_left: int, clip_right: int, want_ops: bool = True, ops_stride: int = 0):
        n = len(tasks)
        arr = (DpTask * max(n, 1))(*tasks)
        res = (DpResult * max(n, 1))()
        ops = None
        if want_ops:
            if not ops_stride:
                ops_stride = max(t.win_len for t in tasks) + 400
            ops = np.zeros(n * ops_stride, dtype=np.uint8)
        f, t = C.c_float(), C.c_float()
        _chk(load_library().mp_sgdp_batch(self._h, arr, n, clip_left, clip_right, res, ops.ctypes.data if want_ops else None,
                                          ops_stride, C.byref(f), C.byref(t)), "mp_sgdp_batch")
        return res, ops, ops_stride, (f.value, t.value)

    def rescue_round(self, params: AlignParams, skip_default_dp: bool = False, fetch: bool = True):
        lib = load_library()
        st = RescueStats()
        _chk(lib.mp_rescue_round(self._h, C.byref(params), 1 if skip_default_dp else 0, C.byref(st)), "mp_rescue_round")
        if not fetch:
            return None, None, st
        return self.single_hits_view(), self.rescue_hits_view(), st

    def single_hits_view(self):
        """(hits, n): the passing single-end DP instances in the context's pinned buffer (valid until the next round)."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _chk(load_library().mp_get_single_hits_view(self._h, C.byref(ptr), C.byref(n)), "mp_get_single_hits_view")
        return ((SingleHit * n.value).from_address(ptr.value) if n.value else (SingleHit * 1)()), n.value

    def rescue_hits_view(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        _chk(load_library().mp_get_rescue_hits_view(self._h, C.byref(ptr), C.byref(n)), "mp_get_rescue_hits_view")
        return ((RescueHit * n.value).from_address(ptr.value) if n.value else (RescueHit * 1)()), n.value

    # ---- per-reference best scores reduced on the device over the chunks a batch is aligned against (mp_reduce_*, what the executable
    # prints its header comments from): begin, one call per chunk after its two rounds, finish
    def set_hit_copies(self, on: bool):
        """False: the rounds leave their hit lists on the device (the *_view calls then fail): only the reduced list is wanted."""
        _chk(load_library().mp_ctx_set_hit_copies(self._h, int(bool(on))), "mp_ctx_set_hit_copies")

    def reduce_begin(self, n_chunks: int):
        _chk(load_library().mp_reduce_begin(self._h, n_chunks), "mp_reduce_begin")

    def reduce_chunk(self, mode: int, chunk_index: int, pair_class: np.ndarray = None, parts: int = 3):
        """-> (n_deep_pairs, n_rescued_pairs, n_unpaired_pairs, n_entries); pair_class (uint8[n_pairs], optional) is filled.  parts: 1 = the pair hits of the
        deep-DP round only, 2 = the rescue round's hits + counts + classes, 3 = both (the two passes of the -top rule, mp_reduce_chunk_part)"""
        st = (C.c_uint64 * 4)()
        _chk(load_library().mp_reduce_chunk_part(self._h, mode, chunk_index, parts, st, pair_class.ctypes.data if pair_class is not None else None), "mp_reduce_chunk_part")
        return tuple(int(x) for x in st)

    def set_score_floor(self, top_percent: int):
        """0: every rescue round does all its work; 1 .. 100: pairs that cannot put an entry of top% of their best known score into a header lose their rescue work"""
        _chk(load_library().mp_ctx_set_score_floor(self._h, int(top_percent)), "mp_ctx_set_score_floor")

    def test_floor_bound(self, tasks_np: np.ndarray) -> np.ndarray:
        """test hook: kernels_floor.hip's upper bound on the DP score of every instance (uint16; 0xFFFF = shape not covered)"""
        assert tasks_np.dtype == DP_TASK_DTYPE and tasks_np.flags.c_contiguous
        out = np.zeros(max(len(tasks_np), 1), dtype=np.uint16)
        _chk(load_library().mp_test_floor_bound(self._h, tasks_np.ctypes.data, len(tasks_np), out.ctypes.data), "mp_test_floor_bound")
        return out[:len(tasks_np)]

    def test_floor_mate_bound(self, tasks_np: np.ndarray, insert_high: int) -> np.ndarray:
        """test hook: the bound on the score of the mate (read_id ^ 1) rescued from a hit inside each instance's window (tasks grouped by read, ascending)"""
        assert tasks_np.dtype == DP_TASK_DTYPE and tasks_np.flags.c_contiguous
        out = np.zeros(max(len(tasks_np), 1), dtype=np.uint16)
        _chk(load_library().mp_test_floor_mate_bound(self._h, tasks_np.ctypes.data, len(tasks_np), int(insert_high), out.ctypes.data), "mp_test_floor_mate_bound")
        return out[:len(tasks_np)]

    def chunk_stash(self, slot: int):
        _chk(load_library().mp_chunk_stash(self._h, slot), "mp_chunk_stash")

    def chunk_unstash(self, slot: int):
        _chk(load_library().mp_chunk_unstash(self._h, slot), "mp_chunk_unstash")

    def reduce_finish(self) -> np.ndarray:
        """the batch's keys (read << 42 | chunks after << 32 | sequence << 10 | 1023 - score), sorted, best score per (read, chunk, sequence):
        a view of the context's pinned buffer, valid until the next reduce_begin"""
        ptr, n = C.c_void_p(), C.c_uint64()
        _chk(load_library().mp_reduce_finish(self._h, C.byref(ptr), C.byref(n)), "mp_reduce_finish")
        if not n.value:
            return np.zeros(0, dtype=np.uint64)
        return np.frombuffer((C.c_uint64 * n.value).from_address(ptr.value), dtype=np.uint64)

    def deep_dp_round(self, params: AlignParams, fetch: bool = True):
        nh = C.c_uint64()
        st = RoundStats()
        _chk(load_library().mp_deep_dp_round(self._h, C.byref(params), C.byref(nh), C.byref(st)), "mp_deep_dp_round")
        if not fetch:                      # the hits are on their way to the host; pair_hits_view() waits for them
            return None, nh.value, st
        return self.pair_hits_view() + (st,)

    def pair_hits_view(self):
        """(hits, n): view of the context's pinned staging buffer -- one DMA started by the round itself, no pageable copy;
        valid until the next round."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _chk(load_library().mp_get_pair_hits_view(self._h, C.byref(ptr), C.byref(n)), "mp_get_pair_hits_view")
        hits = (PairHit * max(n.value, 1)).from_address(ptr.value) if n.value else (PairHit * 1)()
        return hits, n.value
