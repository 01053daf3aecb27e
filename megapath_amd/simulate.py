"""Vectorised synthetic workloads on the GPU (torch is plumbing here: RNG, gathers, packing).

genome(): uniform ACGT with copied repeats (SURVEY.md section 8(d), config C2 shape);
paired_reads(): FR pairs drawn from the genome with substitutions and at most one short indel per mate, plus a share
of unmappable (random) pairs; returns 2-bit packed reads already laid out as mp_set_reads_device expects.
"""
from __future__ import annotations

import torch


def genome(n_bases: int, seed: int, device, repeat_frac: float = 0.10) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    text = torch.randint(0, 4, (n_bases,), dtype=torch.uint8, device=device, generator=g)
    if repeat_frac > 0:
        cpu = torch.Generator()
        cpu.manual_seed(seed + 1)
        done, target = 0, int(n_bases * repeat_frac)
        while done < target:
            ln = int(torch.randint(1000, 5000, (1,), generator=cpu))
            src = int(torch.randint(0, n_bases - ln, (1,), generator=cpu))
            copies = int(torch.randint(2, 50, (1,), generator=cpu))
            dst = torch.randint(0, n_bases - ln, (copies,), generator=cpu)
            seg = text[src:src + ln].clone()
            for d in dst.tolist():
                text[d:d + ln] = seg
            done += ln * copies
    return text


def _mutated_read(text, start, length, g, sub, indel_frac, device):
    """start: int64 [n]; returns codes [n, length] taken from text[start ...] with substitutions and <= 1 indel."""
    n = start.numel()
    i = torch.arange(length, device=device, dtype=torch.int64).unsqueeze(0)          # [1, L]
    has = torch.rand(n, device=device, generator=g) < indel_frac
    is_del = torch.rand(n, device=device, generator=g) < 0.5
    p = torch.randint(10, max(11, length - 10), (n,), device=device, generator=g).unsqueeze(1)
    glen = torch.randint(1, 4, (n,), device=device, generator=g).unsqueeze(1)
    dele = (has & is_del).unsqueeze(1)
    ins = (has & ~is_del).unsqueeze(1)
    # deletion: read skips glen reference bases after p;  insertion: glen random bases at p, the rest shifts back
    off = torch.where(dele & (i >= p), glen, torch.zeros_like(glen))
    off = off - torch.where(ins & (i >= p + glen), glen, torch.zeros_like(glen))
    idx = (start.unsqueeze(1) + i + off).clamp_(0, text.numel() - 1)
    codes = text[idx]
    rnd = torch.randint(0, 4, (n, length), dtype=torch.uint8, device=device, generator=g)
    inserted = ins & (i >= p) & (i < p + glen)
    codes = torch.where(inserted, rnd, codes)
    subm = torch.rand(n, length, device=device, generator=g) < sub          # sub: a rate or one rate per row ([n, 1])
    codes = torch.where(subm, (codes + 1 + (rnd % 3)) & 3, codes)
    return codes.to(torch.uint8)


def _revcomp_prefix(fwd: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """Row r: reverse complement of fwd[r, :lens[r]], left-aligned, zero padded."""
    n, L = fwd.shape
    j = torch.arange(L, device=fwd.device, dtype=torch.int64).unsqueeze(0)
    src = (lens.unsqueeze(1) - 1 - j).clamp_(min=0)
    out = 3 - torch.gather(fwd, 1, src)
    return torch.where(j < lens.unsqueeze(1), out, torch.zeros_like(out))


def paired_reads(text: torch.Tensor, n_pairs: int, read_len: int, seed: int, frag_mean: float = 350.0, frag_sd: float = 40.0,
                 sub: float = 0.01, indel_frac: float = 0.12, random_frac: float = 0.05, sub_max=None, len_range=None,
                 broken_mate_frac: float = 0.0, frag_max: int = 600, return_truth: bool = False):
    """Returns (codes [2*n_pairs, Lmax] uint8, lengths int32 [2*n_pairs]); rows 2p / 2p+1 are the mates of pair p.

    sub_max: per-pair substitution rate drawn uniformly from [0, sub_max] instead of the fixed `sub` (config C3: 0-8 %
    divergence); len_range=(lo, hi): every mate's length uniform in [lo, hi] (config C5; rows are zero padded to hi);
    broken_mate_frac: share of pairs with one mate replaced by random sequence (config C5)."""
    device = text.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = text.numel()
    Lmax = read_len if len_range is None else int(len_range[1])
    if len_range is None:
        l1 = torch.full((n_pairs,), read_len, dtype=torch.int64, device=device)
        l2 = l1.clone()
    else:
        l1 = torch.randint(int(len_range[0]), int(len_range[1]) + 1, (n_pairs,), device=device, generator=g)
        l2 = torch.randint(int(len_range[0]), int(len_range[1]) + 1, (n_pairs,), device=device, generator=g)
    flen = (torch.randn(n_pairs, device=device, generator=g) * frag_sd + frag_mean).to(torch.int64)
    flen = torch.maximum(flen, torch.maximum(l1, l2) + 8).clamp_(max=frag_max)
    pos = (torch.rand(n_pairs, device=device, generator=g) * (n - frag_max - Lmax - 100)).to(torch.int64)
    if sub_max is not None:
        sub = (torch.rand(n_pairs, device=device, generator=g) * float(sub_max)).unsqueeze(1)
    left = _mutated_read(text, pos, Lmax, g, sub, indel_frac, device)
    right_fwd = _mutated_read(text, pos + flen - l2, Lmax, g, sub, indel_frac, device)
    col = torch.arange(Lmax, device=device, dtype=torch.int64).unsqueeze(0)
    left = torch.where(col < l1.unsqueeze(1), left, torch.zeros_like(left))
    right = _revcomp_prefix(right_fwd, l2)
    flip = torch.rand(n_pairs, device=device, generator=g) < 0.5
    if len_range is not None:
        # after the flip read 1 has the right mate's length: draw the fragment from the strand instead, lengths stay put
        left_as_2 = _revcomp_prefix(_mutated_read(text, pos + flen - l1, Lmax, g, sub, indel_frac, device), l1)
        right_as_1 = torch.where(col < l2.unsqueeze(1), _mutated_read(text, pos, Lmax, g, sub, indel_frac, device), torch.zeros_like(left))
        r1 = torch.where(flip.unsqueeze(1), left_as_2, left)
        r2 = torch.where(flip.unsqueeze(1), right_as_1, right)
        # flipped pairs: read 1 (length l1) is the reverse-strand mate at the fragment's end, read 2 (length l2) its start
    else:
        r1 = torch.where(flip.unsqueeze(1), right, left)
        r2 = torch.where(flip.unsqueeze(1), left, right)
    rr = torch.randint(0, 4, (2, n_pairs, Lmax), dtype=torch.uint8, device=device, generator=g)
    rnd_pair = torch.rand(n_pairs, device=device, generator=g) < random_frac
    rnd1, rnd2 = rnd_pair.clone(), rnd_pair.clone()
    if broken_mate_frac > 0:
        broken = torch.rand(n_pairs, device=device, generator=g) < broken_mate_frac
        which = torch.rand(n_pairs, device=device, generator=g) < 0.5
        rnd1 |= broken & which
        rnd2 |= broken & ~which
    r1 = torch.where(rnd1.unsqueeze(1) & (col < l1.unsqueeze(1)), rr[0], r1)
    r2 = torch.where(rnd2.unsqueeze(1) & (col < l2.unsqueeze(1)), rr[1], r2)
    codes = torch.stack([r1, r2], dim=1).reshape(2 * n_pairs, Lmax).contiguous()
    lens = torch.stack([l1, l2], dim=1).reshape(2 * n_pairs).to(torch.int32).contiguous()
    if return_truth:
        # where the fragment came from: text position of its first base, its length, and whether read 1 is the reverse-strand mate
        return codes, lens, dict(pos=pos, flen=flen, flip=flip, random1=rnd1, random2=rnd2)
    return codes, lens


def pack_codes(codes: torch.Tensor, words_per_read: int) -> torch.Tensor:
    """codes [n, L] uint8 -> int32 [n, W] holding the 2-bit LSB-first packing of mp_set_reads (bit pattern of uint32)."""
    n, L = codes.shape
    W = words_per_read
    pad = torch.zeros(n, W * 16, dtype=torch.int64, device=codes.device)
    pad[:, :L] = codes.to(torch.int64)
    sh = 2 * torch.arange(16, device=codes.device, dtype=torch.int64)
    words = (pad.view(n, W, 16) << sh).sum(dim=2)
    return words.to(torch.int32).contiguous()        # wraps to the same 32 bits


def codes_to_ascii(codes_row) -> bytes:
    return bytes(b"ACGT"[int(c)] for c in codes_row)
