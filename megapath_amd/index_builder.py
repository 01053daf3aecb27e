"""2BWT index construction (the builder side of 2bwt-lib, off the hot path).

Builds, from sequences of base codes or a FASTA parsed the reference's way, the arrays the reference's `2bwt-builder` writes to disk
(2BWT-Builder.c:199-393, BWTConstruct.c, file layouts in SURVEY App. A): BWT with '$' omitted, inverseSa0, cumulative frequencies,
sampled SA, packed text -- and writes them as reference-format index files (`write_index_files`: .bwt .fmv .sa .pac .ann .amb .tra
[.lkt]), byte-identical to the real builder's (tests/golden/g1, tests/golden/nrun), which the reference soap4 and this repo's loaders
both read; or hands them to the HIP library directly (`mp_index_from_arrays`).

The compute is the HIP library's `mp_build_index_arrays` (kernels_build.hip: suffix array by prefix doubling, device-wide radix sorts
of (64-bit key, suffix) pairs and one streaming kernel per step; texts up to 2^32 - 3 bases) whenever a GPU is there.  `suffix_array`
below is the same algorithm in torch calls: it runs on the CPU, where the not-gpu tests compare it with the real builder's files, and
is limited to 3.03e9 bases (63-bit rank-pair key).  FASTA parsing (HSPParseFASTAToPacked, HSP.c:354-567: runs of >= 10 ambiguity
letters removed and recorded for .tra / .amb, shorter ones replaced) and the file writers are host code.
"""
from __future__ import annotations

import os
import re

import numpy as np
import torch


last_build_info = None      # {rounds, ms_suffix_sort} of the most recent native build


def _device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


_CH = 1 << 30          # this torch build mishandles several ops beyond INT_MAX elements: they are applied in slices


def _slices(n):
    return [(c, min(n, c + _CH)) for c in range(0, n, _CH)]


def _gather(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    if idx.numel() < 2 ** 31 - 1:
        return src[idx]
    out = torch.empty(idx.numel(), dtype=src.dtype, device=src.device)
    for c0, c1 in _slices(idx.numel()):
        out[c0:c1] = src[idx[c0:c1]]
    return out


def _scatter(dst: torch.Tensor, idx: torch.Tensor, val: torch.Tensor) -> None:
    for c0, c1 in _slices(idx.numel()):
        dst[idx[c0:c1]] = val[c0:c1]


def _cumsum(x: torch.Tensor) -> torch.Tensor:
    if x.numel() < 2 ** 31 - 1:
        return torch.cumsum(x, 0)
    out = torch.empty_like(x)
    carry = 0
    for c0, c1 in _slices(x.numel()):
        out[c0:c1] = torch.cumsum(x[c0:c1], 0) + carry
        carry = int(out[c1 - 1].item())
    return out


def _nonzero(mask_fn, n, device) -> torch.Tensor:
    """Indices where mask_fn(c0, c1) (a bool tensor for that slice) holds."""
    parts = [torch.nonzero(mask_fn(c0, c1)).squeeze(1) + c0 for c0, c1 in _slices(n)]
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=device)


def _argsort_big(key: torch.Tensor) -> torch.Tensor:
    """argsort for more than INT_MAX elements (torch.sort's limit): the keys are cut into value ranges that are sorted
    one by one -- ranges of the key are ranges of the order."""
    n = key.numel()
    if n < 2 ** 31 - 1:
        return torch.argsort(key, stable=False)
    parts = 8
    while True:
        kmax = max(int(key[c0:c1].max().item()) for c0, c1 in _slices(n)) + 1
        step = (kmax + parts - 1) // parts
        sizes = [0] * parts
        for c0, c1 in _slices(n):
            cnt = torch.bincount(torch.div(key[c0:c1], step, rounding_mode="floor"), minlength=parts).cpu().tolist()
            sizes = [a + b for a, b in zip(sizes, cnt)]
        if max(sizes) < 2 ** 31 - 1:
            break
        parts *= 2
    order = torch.empty(n, dtype=torch.int64, device=key.device)
    off = 0
    for b in range(parts):
        if sizes[b] == 0:
            continue
        lo, hi = b * step, (b + 1) * step
        idx = _nonzero(lambda c0, c1: (key[c0:c1] >= lo) & (key[c0:c1] < hi), n, key.device)
        sub = torch.argsort(key[idx], stable=False)
        order[off:off + sizes[b]] = idx[sub]
        off += sizes[b]
        del idx, sub
    return order


def suffix_array(codes: torch.Tensor) -> torch.Tensor:
    """Suffix array of codes + '$' ('$' smallest), length n+1, int64, on codes.device.  Prefix doubling."""
    dev = codes.device
    n = codes.numel()
    assert (n + 2) * (n + 2) < 2 ** 63 and n + 2 < 2 ** 32 - 16      # the (rank, next rank) key must fit 63 bits; index chunks are < 2^32
    K = 20                                              # 5^20 < 2^63: first pass ranks by 20-mers
    sym = torch.zeros(n + 1 + K, dtype=torch.int64, device=dev)
    sym[:n] = codes.to(torch.int64) + 1                 # '$' and everything past the end = 0
    key = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    for t in range(K):
        key = key * 5 + sym[t:t + n + 1]
    del sym
    h = K
    while True:
        order = _argsort_big(key)
        sk = _gather(key, order)
        newgrp = torch.ones(n + 1, dtype=torch.int64, device=dev)
        newgrp[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
        rank_sorted = _cumsum(newgrp) - 1
        del sk, newgrp
        rank = torch.empty(n + 1, dtype=torch.int64, device=dev)
        _scatter(rank, order, rank_sorted)
        done = int(rank_sorted[-1].item()) == n
        del rank_sorted
        if done:
            return order
        del order
        nxt = torch.zeros(n + 1, dtype=torch.int64, device=dev)      # rank of suffix i+h, 0 when it runs off the end
        if h <= n:
            nxt[: n + 1 - h] = rank[h:] + 1
        key = rank * (n + 2) + nxt
        del rank, nxt
        h *= 2


# ---- FASTA parsing as the reference builder does it (HSPParseFASTAToPacked, 2bwt-lib/HSP.c:354-567)
_WS = re.compile(rb"[ \t\n\v\f\r]")
_DNA_CHAR = b"ACGTMRSVWYHKDBN"        # dnaChar[0..14] (HSP.h:154); 'L' = nonMatchDnaCharIndex and everything else is skipped


def parse_fasta_reference_style(data: bytes):
    """-> dict(names, gi, seqs [uint8 codes 0..3], ambiguity [(startPos, rightOfEndPos)], first_amb_index [per sequence],
    actual [(startPos, length) per sequence incl. removed runs]).  Rules (HSP.c:441-547): the name is the header up to the first
    white space; letters outside IUPAC are skipped (so is 'U'); lower case counts as upper case; a run of ambiguity letters
    (anything but ACGT, white space inside it ignored) shorter than 10 becomes that many 'G', a run of 10 or more is REMOVED from
    the packed text and recorded; a '>' in the sequence body starts the next record, a '>' inside a header line does not."""
    kind = np.zeros(256, dtype=np.uint8)                 # 0 skip, 1..4 = A C G T (code + 1), 5 = ambiguity letter
    for i, ch in enumerate(_DNA_CHAR):
        v = i + 1 if i < 4 else 5
        kind[ch] = v
        kind[ch + 32] = v
    buf = np.frombuffer(data, dtype=np.uint8)
    if len(buf) == 0 or buf[0] != ord(">"):
        raise ValueError("FASTA file does not begin with '>'")
    names, gis, seqs, ambiguity, first_amb, actual = [], [], [], [], [], []
    total, discarded = 0, 0
    h0 = 0
    while h0 < len(buf):
        # the header runs to the end of its line whatever it contains ("3'->5' exonuclease"); only then does a '>' start the next
        # record (HSP.c:441-449 reads to '\n', :476 scans the body for '>')
        nl = data.find(b"\n", h0)
        if nl < 0:
            nl = len(buf)                                 # header without a newline: a record with an empty sequence
        rec_end = data.find(b">", nl + 1) if nl + 1 < len(buf) else -1
        if rec_end < 0:
            rec_end = len(buf)
        header = data[h0 + 1:nl]
        m = _WS.search(header)                            # the name ends at the first white-space byte (a header that begins
        name = header[:m.start()] if m else header        # with one has an empty name, HSP.c:442-445)
        name = name[:255]                                 # MAX_SEQ_NAME_LENGTH
        gi = 0
        if len(name) > 3 and name[:3] == b"gi|":
            digits = b""
            for ch in name[3:]:
                if 48 <= ch <= 57:
                    digits += bytes([ch])
                else:
                    break
            gi = int(digits) if digits else 0
        k = kind[buf[nl + 1:rec_end]] if nl + 1 < rec_end else np.zeros(0, dtype=np.uint8)
        k = k[k != 0]
        amb = k == 5
        codes = (k - 1).astype(np.uint8)
        start_total, start_actual = total, total + discarded
        keep = np.ones(len(k), dtype=bool)
        if amb.any():
            d = np.diff(np.concatenate([[0], amb.view(np.int8), [0]]))
            run_s, run_e = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
            removed_before = 0
            for s0, e0 in zip(run_s, run_e):
                ln = int(e0 - s0)
                if ln < 10:
                    codes[s0:e0] = 2                      # 'G'
                else:
                    keep[s0:e0] = False
                    discarded += ln
                    pos = total + int(s0) - removed_before
                    ambiguity.append((pos, pos + discarded - 1))
                    removed_before += ln
        codes = codes[keep]
        names.append(name.decode())
        gis.append(gi)
        seqs.append(codes)
        total += len(codes)
        first_amb.append(len(ambiguity))
        actual.append((start_actual, total + discarded - start_actual))
        h0 = rec_end
    return dict(names=names, gi=gis, seqs=seqs, ambiguity=ambiguity, first_amb_index=first_amb, actual=actual)


def translate_table(n, starts, ambiguity, first_amb_index):
    """The .tra body as HSP.c:569-640 computes it: grid (one entry per 2^18 packed bases) and the merged list of removed-run
    and sequence-start rows with their coordinate corrections, in 64-bit unsigned arithmetic, quirks included (a sequence start
    that coincides with a removed run gets no row of its own; the list is padded with copies of its last row).  One read of
    the reference is out of bounds (ambiguity[-1] when no run precedes sequence k-2, :585-586): the value it finds there made the
    term 1 in every build tried, which is what is used here."""
    M = (1 << 64) - 1
    n_seq, n_amb = len(starts), len(ambiguity)
    grid_entries = (n >> 18) + 1
    grid = [0] * grid_entries
    rows = []

    def amb_row(j, k):
        a0, a1 = ambiguity[j]
        if k - 1 > 0:
            q = first_amb_index[k - 2] - 1
            corr = (ambiguity[q][1] - ambiguity[q][0] + 1) if q >= 0 else 1
            c = (int(starts[k - 1]) + corr - (a1 - a0 + 1) - 1) & M
        else:
            c = (-a1 + a0 - 2) & M
        grid[a0 >> 18] += 1
        rows.append((a0, k, c))
    j = k = 0
    while j < n_amb and k < n_seq:
        if ambiguity[j][0] < int(starts[k]):
            amb_row(j, k)
            j += 1
        elif ambiguity[j][0] > int(starts[k]):
            grid[int(starts[k]) >> 18] += 1
            rows.append((int(starts[k]), k + 1, (int(starts[k]) - 1) & M))
            k += 1
        else:
            k += 1
    while j < n_amb:
        amb_row(j, k)
        j += 1
    while k < n_seq:
        grid[int(starts[k]) >> 18] += 1
        rows.append((int(starts[k]), k + 1, (int(starts[k]) - 1) & M))
        k += 1
    while len(rows) < n_amb + n_seq:
        rows.append(rows[-1] if rows else (0, 0, 0))
    acc, out_grid = 0, []
    for g in grid:
        acc += g
        out_grid.append((acc - 1) & 0xFFFFFFFF)
    return out_grid, rows


def build_index_arrays(seqs, sa_interval: int = 16, device=None, native=None):
    """seqs: list of (name, uint8 array of base codes 0..3 or ASCII letters), or the dict parse_fasta_reference_style() returns
    (then runs of >= 10 ambiguity letters are already removed and recorded for .tra).  Returns the keyword arguments of
    `megapath_amd.hipapi.Index.from_arrays` (host numpy arrays in the reference's file layouts) plus what write_index_files needs.

    native: True = the HIP builder (mp_build_index_arrays: prefix doubling with device-wide radix sorts and its own kernels), False = the
    torch restatement below (runs on the CPU too: the not-gpu tests compare it with the real builder's files); None = native when a GPU
    is there.  Both give identical arrays (tests/test_gpu_parity.py)."""
    if native is None:
        native = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
    dev = device or _device()
    parsed = None
    if isinstance(seqs, dict):
        parsed = seqs
        seqs = list(zip(parsed["names"], parsed["seqs"]))
    names, starts, lengths, parts = [], [], [], []
    pos = 0
    for name, arr in seqs:
        a = np.asarray(arr, dtype=np.uint8)
        if a.size and a.max() > 3:                       # ASCII
            lut = np.zeros(256, dtype=np.uint8)
            for i, ch in enumerate(b"ACGT"):
                lut[ch] = i
                lut[ch + 32] = i
            lut[ord("U")] = lut[ord("u")] = 3
            lut[ord("N")] = lut[ord("n")] = 2
            a = lut[a]
        names.append(name)
        starts.append(pos)
        lengths.append(len(a))
        parts.append(a)
        pos += len(a)
    text = np.concatenate(parts)
    n = len(text)
    if native:
        from . import hipapi
        dev_index = torch.device(device).index if device is not None and torch.device(device).index is not None else \
            (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        core, info = hipapi.build_index_arrays_native(text, sa_interval, dev_index)
        out = dict(core)
        out["pac"] = np.concatenate([out["pac"], np.zeros(8, dtype=np.uint8)])
        out.update(sa_interval=sa_interval, names=names, starts=np.asarray(starts, dtype=np.uint64), lengths=np.asarray(lengths, dtype=np.uint64))
        if parsed is not None:
            out["_fasta"] = dict(gi=parsed["gi"], ambiguity=parsed["ambiguity"], first_amb_index=parsed["first_amb_index"], actual=parsed["actual"])
        global last_build_info
        last_build_info = info
        return out
    codes = torch.from_numpy(text).to(dev)
    sa = suffix_array(codes)                            # n+1 entries, sa[0] == n
    inv_sa0 = int(_nonzero(lambda c0, c1: sa[c0:c1] == 0, n + 1, dev)[0].item())
    # BWT with the '$' row removed (BWT.c:605): rows in order, the one whose suffix starts the text dropped
    prev = torch.cat([sa[:inv_sa0], sa[inv_sa0 + 1:]]) - 1
    bwt = _gather(codes, prev).to(torch.uint8)
    del prev
    cnt = torch.bincount(codes.to(torch.int64), minlength=4).cpu().numpy()
    cum = np.zeros(5, dtype=np.uint64)
    cum[1:] = np.cumsum(cnt)
    # 16 symbols per word, MSB first
    nb = (n + 15) // 16 * 16
    pad = torch.zeros(nb, dtype=torch.int64, device=dev)
    pad[:n] = bwt.to(torch.int64)
    shifts = (2 * (15 - torch.arange(16, device=dev, dtype=torch.int64)))
    bwt_words = (pad.view(-1, 16) << shifts).sum(dim=1).to(torch.int64).cpu().numpy().astype(np.uint32)
    del pad, bwt
    samples = sa[::sa_interval].cpu().numpy().astype(np.uint64)
    del sa
    # .pac: 4 bases per byte, first base in bits 7-6
    nb4 = (n + 3) // 4 * 4
    p4 = torch.zeros(nb4, dtype=torch.int64, device=dev)
    p4[:n] = codes.to(torch.int64)
    sh4 = (2 * (3 - torch.arange(4, device=dev, dtype=torch.int64)))
    pac = (p4.view(-1, 4) << sh4).sum(dim=1).to(torch.uint8).cpu().numpy()
    pac = np.concatenate([pac, np.zeros(8, dtype=np.uint8)])
    out = dict(n=n, inv_sa0=inv_sa0, cum=cum, bwt_words=bwt_words, sa_samples=samples, sa_interval=sa_interval, pac=pac,
               names=names, starts=np.asarray(starts, dtype=np.uint64), lengths=np.asarray(lengths, dtype=np.uint64))
    if parsed is not None:
        out["_fasta"] = dict(gi=parsed["gi"], ambiguity=parsed["ambiguity"], first_amb_index=parsed["first_amb_index"], actual=parsed["actual"])
    return out


def index_kwargs(arrs):
    """The subset of build_index_arrays' result that hipapi.Index.from_arrays takes."""
    return {k: v for k, v in arrs.items() if not k.startswith("_")}


def write_index_files(prefix: str, arrs, write_lkt: bool = False, lkt_table=None) -> None:
    """Write <prefix>.{bwt,fmv,sa,pac,ann,amb,tra} in the reference's formats so that the reference
    soap4, the oracle and mp_index_open can all load the index.  The 512 MiB .lkt is optional (every loader in this
    repo rebuilds it from .pac; the reference soap4 needs the file)."""
    n = int(arrs["n"])
    hdr = np.zeros(5, dtype=np.uint64)
    hdr[0] = arrs["inv_sa0"]
    hdr[1:] = arrs["cum"][1:]
    with open(prefix + ".bwt", "wb") as f:
        f.write(hdr.tobytes())
        f.write(np.asarray(arrs["bwt_words"], dtype=np.uint32).tobytes())
    # explicit occurrences (BWTConstruct.c:1229-1233; read side BWT.c:783-797): symbol counts per 256-symbol interval from
    # a per-byte table (4 symbols per byte), never materialising one element per base
    bw = np.asarray(arrs["bwt_words"], dtype=np.uint32)
    n_occ = (n + 255) // 256 + 1
    # the reference counts whole 256-symbol intervals after clearing the trailing code (BWTConstruct.c:312-314, 1036-1041):
    # the sample past the text end includes the zero padding as 'A's, and BWTOccValue subtracts them again when it
    # decodes backwards from that sample (reachable whenever n mod 256 > 128)
    words = np.zeros((n_occ - 1) * 16, dtype=np.uint32)
    words[: len(bw)] = bw
    if n % 16:
        words[n // 16] &= np.uint32(0xFFFFFFFF) << np.uint32(2 * (16 - n % 16))          # clear the trailing code (MSB-first symbols)
    words[(n + 15) // 16:] = 0
    by = words.view(np.uint8)
    tab = np.zeros((256, 4), dtype=np.uint8)
    for v in range(256):
        for k in range(4):
            tab[v, (v >> (2 * k)) & 3] += 1
    occ = np.zeros((n_occ, 4), dtype=np.uint64)
    for c in range(4):
        per_iv = tab[by, c].reshape(n_occ - 1, 64).sum(axis=1, dtype=np.uint64)
        occ[1:, c] = np.cumsum(per_iv, dtype=np.uint64)
    n_major = (n_occ + 255) // 256
    major = occ[::256][:n_major].copy()
    rel = (occ - np.repeat(major, 256, axis=0)[:n_occ]).astype(np.uint32)
    n_pairs = (n_occ + 1) // 2
    minor = np.zeros((n_pairs, 4), dtype=np.uint32)
    relp = np.zeros((n_pairs * 2, 4), dtype=np.uint32)
    relp[:n_occ] = rel
    minor[:] = (relp[0::2] << 16) + (relp[1::2] & 0xFFFF)
    with open(prefix + ".fmv", "wb") as f:
        f.write(hdr.tobytes())
        f.write(minor.tobytes())
        f.write(major.astype(np.uint64).tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(hdr.tobytes())
        f.write(np.uint64(arrs["sa_interval"]).tobytes())
        f.write(np.asarray(arrs["sa_samples"], dtype=np.uint64).tobytes())
    pac = np.asarray(arrs["pac"], dtype=np.uint8)[: (n + 3) // 4]
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.tobytes())
        if n % 4 == 0:
            f.write(b"\0")
        f.write(bytes([n % 4]))
    names, starts, lengths = arrs["names"], arrs["starts"], arrs["lengths"]
    fasta = arrs.get("_fasta") or dict(gi=[0] * len(names), ambiguity=[], first_amb_index=[0] * len(names),
                                      actual=[(int(s), int(l)) for s, l in zip(starts, lengths)])
    with open(prefix + ".ann", "w") as f:                                   # HSP.c:653-657 (the third number is the builder's random seed)
        f.write("%d %d %d\n" % (n, len(names), 11))
        for nm, gi, s, l in zip(names, fasta["gi"], starts, lengths):
            f.write("%d %s\n%d %d 0\n" % (gi, nm, int(s), int(l)))
    with open(prefix + ".amb", "w") as f:                                   # HSP.c:670: the list itself is disabled
        f.write("%d %d 0\n" % (n, len(names)))
    grid, rows = translate_table(n, starts, fasta["ambiguity"], fasta["first_amb_index"])
    with open(prefix + ".tra", "w") as f:                                   # HSP.c:679-690
        f.write("%d %d %d %d\n" % (n, len(names), len(fasta["ambiguity"]), len(grid)))
        for g in grid:
            f.write("%d\n" % g)
        for a, c, corr in rows:
            f.write("%d %d %d\n" % (a, c, corr))
        for s, l in fasta["actual"]:
            f.write("%d %d\n" % (int(s), int(l)))
    if lkt_table is not None:
        # the 13-mer table as a loader already built it (mp_index_copy_lkt): inclusive right bounds, written as u64
        with open(prefix + ".lkt", "wb") as f:
            f.write(np.int32(13).tobytes())
            f.write(np.asarray(lkt_table).astype(np.uint64).tobytes())
    elif write_lkt:
        text = np.concatenate([((pac[:, None] >> (2 * (3 - np.arange(4, dtype=np.uint8)))) & 3).reshape(-1)[:n],
                               np.zeros(12, dtype=np.uint8)]).astype(np.uint64)
        key = np.zeros(n, dtype=np.uint64)
        for t in range(13):
            key = (key << np.uint64(2)) | text[t:t + n]
        cnt = np.bincount(key.astype(np.int64), minlength=4 ** 13).astype(np.uint64)
        table = np.cumsum(cnt, dtype=np.uint64)
        with open(prefix + ".lkt", "wb") as f:
            f.write(np.int32(13).tobytes())
            f.write(table.tobytes())


def main(argv=None) -> int:
    """`2bwt-builder <sequence file>` (2BWT-Builder.c:199-393): writes <file>.index.{bwt,fmv,sa,pac,ann,amb,tra,lkt} like the reference's
    builder -- same parsing of the FASTA (N runs, IUPAC codes, gi| headers), same files (the .ann carries a fixed number where the
    reference writes a time-based seed).  The suffix array is built on the GPU (mp_build_index_arrays); the 13-mer table comes from the
    HIP library's own loader of the new index, so the library has read the files' content once before they are written."""
    import argparse
    import sys
    import time
    ap = argparse.ArgumentParser(prog="2bwt-builder", description="build a 2BWT index (reference-compatible files) on an MI355X")
    ap.add_argument("fasta")
    ap.add_argument("--sa-interval", type=int, default=16, help="SaValueFreq of 2bwt-builder.ini (default 16, what soap4 expects)")
    ap.add_argument("--device", type=int, default=int(os.environ.get("MP_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    ap.add_argument("--no-lkt", action="store_true", help="skip the 512 MiB 13-mer table (this repo's loaders rebuild it; the reference soap4 needs it)")
    args = ap.parse_args(argv)
    from . import hipapi
    t0 = time.time()
    try:
        data = open(args.fasta, "rb").read()
    except OSError as e:
        print("Cannot open sequence file %s: %s" % (args.fasta, e), file=sys.stderr)
        return 1
    parsed = parse_fasta_reference_style(data)
    if not parsed["names"]:
        print("No sequence in %s" % args.fasta, file=sys.stderr)
        return 1
    arrs = build_index_arrays(parsed, sa_interval=args.sa_interval, device="cuda:%d" % args.device, native=True)
    t1 = time.time()
    lkt = None
    if not args.no_lkt:
        gix = hipapi.Index.from_arrays(args.device, **index_kwargs(arrs))
        lkt = gix.lkt_table()
        gix.close()
    write_index_files(args.fasta + ".index", arrs, lkt_table=lkt)
    info = last_build_info or {}
    print("[2bwt-builder] %d bases in %d sequences, %d removed ambiguity runs: suffix array %.2f s (%s doubling rounds), arrays %.1f s, files %.1f s"
          % (arrs["n"], len(arrs["names"]), len(parsed["ambiguity"]), info.get("ms_suffix_sort", 0.0) / 1e3, info.get("rounds", "?"), t1 - t0, time.time() - t1),
          file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
