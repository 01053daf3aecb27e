"""The only cross-GPU exchange of the path: per-reference hit counters that feed `cc/reassign`
pass 1 (reference cc/reassign.cpp:24-25,93-104: counts[taxid], uniq_counts[taxid]).  Every rank aligns its own shard of
the pairs against its own replica of the index; what has to be global is only, per reference sequence, how many reads
hit it (best hits) and how many hit it uniquely.  Dense int64 vectors indexed by sequence id, one all-reduce(sum) --
RCCL over xGMI on GPUs (`backend="nccl"`), gloo in the CPU tests.  Payload = 16 B per reference sequence: latency-bound.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def counts_from_fastq(lines, seq_index: dict, n_seqs: int):
    """Accumulate counters from annotated-FASTQ header lines (`@name\\tSCORE:<best>;<score>,<ref>;...`), the stream
    `cc/fastq2lsam` consumes (cc/fastq2lsam.cpp:28-78).  counts[ref] += 1 for every listed reference of a read,
    uniq[ref] += 1 when it is the only one."""
    counts = np.zeros(n_seqs, dtype=np.int64)
    uniq = np.zeros(n_seqs, dtype=np.int64)
    for h in lines:
        if not h.startswith(b"@"):
            continue
        tab = h.find(b"\tSCORE:")
        if tab < 0:
            continue
        fields = [f for f in h[tab + 7:].rstrip(b"\n").split(b";")[1:] if f]
        refs = [seq_index[f.split(b",", 1)[1].decode()] for f in fields if b"," in f]
        for r in refs:
            counts[r] += 1
        if len(refs) == 1:
            uniq[refs[0]] += 1
    return counts, uniq


def all_reduce_counts(counts: np.ndarray, uniq: np.ndarray, device=None):
    """Sum the per-rank counters over the process group; returns numpy arrays identical on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return counts.copy(), uniq.copy()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.from_numpy(np.stack([counts, uniq]).astype(np.int64)).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy()
    return out[0], out[1]


PAIR_HIT_DTYPE = np.dtype([("pair_id", "<u4"), ("left_is_mate", "<u4"), ("algnmt_1", "<u8"), ("algnmt_2", "<u8"), ("score_1", "<i4"), ("score_2", "<i4"),
                           ("start_1", "<u4"), ("start_2", "<u4"), ("dplen_1", "<u2"), ("dplen_2", "<u2"), ("count_1", "<u2"), ("count_2", "<u2")])   # mp_pair_hit


def count_pair_hits_per_reference(index, hits, n_hits: int) -> np.ndarray:
    """int64[num_seq]: deep-DP pair hits per reference sequence (by the left mate's position) of one batch -- the dense
    per-rank vector of the exchange step."""
    n_seq = int(index.info.num_seq)
    if n_hits == 0:
        return np.zeros(n_seq, dtype=np.int64)
    import ctypes as C
    arr = np.frombuffer((C.c_char * (n_hits * PAIR_HIT_DTYPE.itemsize)).from_address(C.addressof(hits)), dtype=PAIR_HIT_DTYPE, count=n_hits)
    _, chrom = index.chr_and_pos_batch(np.ascontiguousarray(arr["algnmt_1"]))
    return np.bincount(chrom.astype(np.int64) - 1, minlength=n_seq).astype(np.int64)


def allreduce_counts(counts: np.ndarray, device) -> np.ndarray:
    """Sum of a dense per-rank counter vector over the process group (RCCL when the group's backend is nccl)."""
    t = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def shard_bounds(n_pairs: int, world: int, rank: int):
    """Contiguous block of pairs of one rank (keeps outputs mergeable in input order)."""
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
