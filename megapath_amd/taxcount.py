"""The one exchange step of the path: the counters of `cc/reassign` pass 1, all-reduced across ranks.

Every rank aligns its own contiguous block of the pairs against its own replica of the index and ends up with its own
shard of the LSAM.id stream (bin/soap4 ... -lsamid ... -pass1 <file>, or the reference's soap4 | fastq2lsam | taxLookupAcc).
`reassign` (reference cc/reassign.cpp) makes two passes over the WHOLE stream: pass 1 (:64-110) counts, per species taxid, the
reads that list it (`counts`), the reads that list only it (`uniq_counts`) and, per pair of taxids, the reads that list both
(`intersect`), keeping reads whose best score reaches `-t`; from those it derives which taxon "explains" which (:112-141); pass 2
(:160-222) drops the explained taxids from every read.  Only the pass-1 counters are global: with them summed over the ranks
-- dense vectors over the union of the ranks' taxids in one all-reduce, the sparse pair table by all-gather -- every rank can run
pass 2 on its own shard and the concatenation equals what `reassign` prints for the unsharded file.  RCCL over xGMI on GPUs
(`backend="nccl"`), gloo in the CPU tests.  Payload: 16 B per taxid + 12 B per co-occurring pair: latency-bound, a few MB at most.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist


@dataclass
class Pass1:
    counts: dict = field(default_factory=dict)        # taxid -> reads listing it
    uniq: dict = field(default_factory=dict)          # taxid -> reads listing only it
    intersect: dict = field(default_factory=dict)     # (larger taxid, smaller taxid) -> reads listing both (pairup(), cc/reassign.cpp:29-32)

    def __eq__(self, o):
        strip = lambda d: {k: v for k, v in d.items() if v}    # noqa: E731
        return strip(self.counts) == strip(o.counts) and strip(self.uniq) == strip(o.uniq) and strip(self.intersect) == strip(o.intersect)


def _hits(field6: bytes):
    """splitAcc on an LSAM.id hit column (cc/misc.h:47-59): [(taxid, score)] in line order."""
    if not field6 or field6 == b"*":
        return []
    out = []
    for el in field6.split(b";"):
        if not el:
            continue
        sc, tid = el.split(b",")[:2]
        out.append((int(tid), float(sc)))
    return out


def pass1_from_lsam_id(lines, t: float = 40.0) -> Pass1:
    """cc/reassign.cpp:64-110 over an iterable of LSAM.id lines (bytes)."""
    p = Pass1()
    for line in lines:
        rec = line.rstrip(b"\n").split(b"\t")
        if len(rec) < 6:
            continue
        hits = _hits(rec[5])
        if float(rec[2]) < t or not hits:
            continue
        seen = []
        for tid, _ in hits:
            p.counts[tid] = p.counts.get(tid, 0) + 1
            if len(hits) == 1:
                p.uniq[tid] = p.uniq.get(tid, 0) + 1
            else:
                for s in seen:
                    k = (max(s, tid), min(s, tid))
                    p.intersect[k] = p.intersect.get(k, 0) + 1
            seen.append(tid)
    return p


def read_pass1_file(path: str) -> Pass1:
    """The counters bin/soap4-mi355x writes with -pass1."""
    p = Pass1()
    for line in open(path):
        f = line.split()
        if not f or f[0].startswith("#"):
            continue
        if f[0] == "counts":
            p.counts[int(f[1])] = int(f[2])
        elif f[0] == "uniq":
            p.uniq[int(f[1])] = int(f[2])
        elif f[0] == "intersect":
            p.intersect[(int(f[1]), int(f[2]))] = int(f[3])
    return p


def all_reduce_pass1(p: Pass1, device=None, always: bool = False) -> Pass1:
    """Sum of the ranks' pass-1 counters; identical on every rank afterwards.  `always`: run the collectives even in a group of one rank
    (the test that executes RCCL on a one-GPU box)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return p
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))

    def gather_var(arr: np.ndarray) -> np.ndarray:
        """all-gather of int64 rows of different counts per rank (padded to the longest)."""
        n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        m = max(int(x.item()) for x in ns)
        pad = torch.zeros((max(m, 1),) + arr.shape[1:], dtype=torch.int64, device=dev)
        if arr.shape[0]:
            pad[:arr.shape[0]] = torch.from_numpy(arr).to(dev)
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        return np.concatenate([b[:int(k.item())].cpu().numpy() for b, k in zip(bufs, ns)], axis=0)

    # 1. the union of the taxids seen anywhere: slot map for the dense vectors
    mine = np.array(sorted(set(p.counts) | set(p.uniq)), dtype=np.int64)
    universe = np.unique(gather_var(mine))
    slot = {int(t): i for i, t in enumerate(universe)}
    dense = torch.zeros((2, max(len(universe), 1)), dtype=torch.int64, device=dev)
    for t, v in p.counts.items():
        dense[0, slot[t]] = v
    for t, v in p.uniq.items():
        dense[1, slot[t]] = v
    dist.all_reduce(dense, op=dist.ReduceOp.SUM)                 # the per-taxon hit-count all-reduce
    dense = dense.cpu().numpy()
    # 2. the sparse pair table: all-gather, merge
    pairs = np.array([[a, b, v] for (a, b), v in sorted(p.intersect.items())], dtype=np.int64).reshape(-1, 3)
    allp = gather_var(pairs)
    out = Pass1()
    for i, t in enumerate(universe):
        if dense[0, i]:
            out.counts[int(t)] = int(dense[0, i])
        if dense[1, i]:
            out.uniq[int(t)] = int(dense[1, i])
    for a, b, v in allp:
        k = (int(a), int(b))
        out.intersect[k] = out.intersect.get(k, 0) + int(v)
    return out


def explains(p: Pass1, u: float = 20.0, v: float = 0.05):
    """cc/reassign.cpp:34-38,112-141: the set of (t1, t2) with "t1 explains t2"."""
    def weakly(t1, t2):
        if p.uniq.get(t1, 0) <= u * p.uniq.get(t2, 0):
            return False
        k = (max(t1, t2), min(t1, t2))
        return not (p.counts.get(t1, 0) - p.intersect.get(k, 0) <= v * p.counts.get(t1, 0))
    weakly_explained = set()
    for (t1, t2) in p.intersect:
        if weakly(t1, t2):
            weakly_explained.add(t2)
        elif weakly(t2, t1):
            weakly_explained.add(t1)
    out = set()
    for (t1, t2) in p.intersect:
        if weakly(t1, t2):
            if t1 not in weakly_explained:
                out.add((t1, t2))
        elif weakly(t2, t1):
            if t2 not in weakly_explained:
                out.add((t2, t1))
    return out


def apply_explains(lines, expl, output_seq: bool = False):
    """Pass 2 of cc/reassign (:160-222) on a shard of LSAM.id lines: a taxid explained by another taxid of the same read goes."""
    out = []
    for line in lines:
        rec = line.rstrip(b"\n").split(b"\t")
        hits = _hits(rec[5])
        if not output_seq:
            rec[3] = rec[4] = b"*"
        kept = [b"%d,%d" % (int(sc), tid) for tid, sc in hits if not any((other, tid) in expl for other, _ in hits)]
        col = b";".join(kept) if hits else b"*"
        out.append(b"\t".join(rec[:5] + [col] + rec[6:]) + b"\n")
    return out


# ------------------------------------------------------------------------------------------------ sharding + bench helpers
PAIR_HIT_DTYPE = np.dtype([("pair_id", "<u4"), ("left_is_mate", "<u4"), ("algnmt_1", "<u8"), ("algnmt_2", "<u8"), ("score_1", "<i4"), ("score_2", "<i4"),
                           ("start_1", "<u4"), ("start_2", "<u4"), ("dplen_1", "<u2"), ("dplen_2", "<u2"), ("count_1", "<u2"), ("count_2", "<u2")])   # mp_pair_hit


def count_pair_hits_per_reference(index, hits, n_hits: int) -> np.ndarray:
    """int64[num_seq]: deep-DP pair hits per reference sequence (by the left mate's position) of one batch -- bench.py's
    stand-in for the per-taxon vector when there is no taxonomy for a synthetic genome."""
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
