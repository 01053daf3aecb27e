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
    subm = torch.rand(n, length, device=device, generator=g) < sub
    codes = torch.where(subm, (codes + 1 + (rnd % 3)) & 3, codes)
    return codes.to(torch.uint8)


def paired_reads(text: torch.Tensor, n_pairs: int, read_len: int, seed: int, frag_mean: float = 350.0, frag_sd: float = 40.0,
                 sub: float = 0.01, indel_frac: float = 0.12, random_frac: float = 0.05):
    """Returns (codes [2*n_pairs, read_len] uint8, lengths int32 [2*n_pairs]); rows 2p / 2p+1 are the mates of pair p."""
    device = text.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = text.numel()
    flen = (torch.randn(n_pairs, device=device, generator=g) * frag_sd + frag_mean).clamp_(read_len + 8, 600).to(torch.int64)
    pos = (torch.rand(n_pairs, device=device, generator=g) * (n - 700)).to(torch.int64)
    left = _mutated_read(text, pos, read_len, g, sub, indel_frac, device)
    right_fwd = _mutated_read(text, pos + flen - read_len, read_len, g, sub, indel_frac, device)
    right = (3 - right_fwd).flip(1)
    flip = torch.rand(n_pairs, device=device, generator=g) < 0.5
    r1 = torch.where(flip.unsqueeze(1), right, left)
    r2 = torch.where(flip.unsqueeze(1), left, right)
    rnd_pair = torch.rand(n_pairs, device=device, generator=g) < random_frac
    rr = torch.randint(0, 4, (2, n_pairs, read_len), dtype=torch.uint8, device=device, generator=g)
    r1 = torch.where(rnd_pair.unsqueeze(1), rr[0], r1)
    r2 = torch.where(rnd_pair.unsqueeze(1), rr[1], r2)
    codes = torch.stack([r1, r2], dim=1).reshape(2 * n_pairs, read_len).contiguous()
    lens = torch.full((2 * n_pairs,), read_len, dtype=torch.int32, device=device)
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
