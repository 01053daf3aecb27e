"""Seeded synthetic genomes and paired-end reads.

Pure numpy; used by tests, bench.py and the fixture generators.  Nothing here is on the hot path.
"""
from __future__ import annotations

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def random_genome(n_bases: int, seed: int, n_seqs: int = 1, repeat_frac: float = 0.0):
    """Return a list of (name, uint8 ASCII array).  Uniform ACGT; optional copied repeats."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cuts = [0] + sorted(rng.choice(np.arange(1, n_bases), size=n_seqs - 1, replace=False).tolist()) + [n_bases] \
        if n_seqs > 1 else [0, n_bases]
    text = _BASES[rng.integers(0, 4, size=n_bases, dtype=np.uint8)]
    if repeat_frac > 0:
        target = int(n_bases * repeat_frac)
        done = 0
        while done < target:
            ln = int(rng.integers(1000, 5000))
            src = int(rng.integers(0, n_bases - ln))
            for _ in range(int(rng.integers(2, 50))):
                dst = int(rng.integers(0, n_bases - ln))
                text[dst:dst + ln] = text[src:src + ln]
                done += ln
                if done >= target:
                    break
    return [("seq%d" % (i + 1), text[cuts[i]:cuts[i + 1]]) for i in range(n_seqs)]


def write_fasta(path: str, seqs, width: int = 70) -> None:
    with open(path, "wb") as f:
        for name, arr in seqs:
            f.write(b">" + name.encode() + b"\n")
            b = arr.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


def _mutate(frag: np.ndarray, rng, sub: float, ins: float, dele: float) -> np.ndarray:
    out = []
    r = rng.random(len(frag) * 2 + 8)
    k = 0
    for ch in frag:
        x = r[k]; k += 1
        if x < dele:
            continue
        if x < dele + ins:
            out.append(_BASES[int(r[k] * 4) & 3]); k += 1
        if x > 1.0 - sub:
            alt = _BASES[(int(np.searchsorted(_BASES, ch)) + 1 + int(r[k] * 3) % 3) & 3]; k += 1
            out.append(alt)
        else:
            out.append(ch)
        if k > len(r) - 4:
            r = rng.random(len(frag) * 2 + 8); k = 0
    return np.asarray(out, dtype=np.uint8)


def simulate_pairs(seqs, n_pairs: int, read_len: int, seed: int, frag_mean: float = 350.0, frag_sd: float = 40.0,
                   frag_min: int = 200, frag_max: int = 600, sub: float = 0.01, ins: float = 0.0005,
                   dele: float = 0.0005, random_frac: float = 0.0, len_range=None, n_frac: float = 0.0,
                   broken_mate_frac: float = 0.0, noisy_mate_frac: float = 0.0, noisy_sub: float = 0.2,
                   long_indel_frac: float = 0.0, long_indel_max: int = 24, junk_end_frac: float = 0.0, junk_end_max: int = 40):
    """Return (names, reads1, reads2) with reads as ASCII byte strings; FR orientation.

    `len_range=(lo,hi)` draws each mate's length uniformly (config C5); `random_frac` replaces whole
    pairs by random sequence; `broken_mate_frac` replaces one mate by random sequence; `noisy_mate_frac` gives one
    mate extra substitutions at rate `noisy_sub` (too few exact seeds: only mate rescue can place it);
    `long_indel_frac` gives a mate one insertion or deletion of 2..`long_indel_max` bases (alignments far off the main
    diagonal of their DP window); `junk_end_frac` replaces up to `junk_end_max` bases at one end of a mate by random
    sequence (soft clips).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.asarray([len(s[1]) for s in seqs], dtype=np.int64)
    prob = lens / lens.sum()
    names, r1s, r2s = [], [], []
    for p in range(n_pairs):
        l1 = l2 = read_len
        if len_range is not None:
            l1 = int(rng.integers(len_range[0], len_range[1] + 1))
            l2 = int(rng.integers(len_range[0], len_range[1] + 1))
        if rng.random() < random_frac:
            a = _BASES[rng.integers(0, 4, size=l1)]
            b = _BASES[rng.integers(0, 4, size=l2)]
        else:
            si = int(rng.choice(len(seqs), p=prob))
            text = seqs[si][1]
            flen = int(min(max(rng.normal(frag_mean, frag_sd), max(frag_min, l1, l2) + 8), frag_max))
            flen = min(flen, len(text) - 1)
            pos = int(rng.integers(0, len(text) - flen))
            frag = _mutate(text[pos:pos + flen], rng, sub, ins, dele)
            if len(frag) < max(l1, l2):
                frag = np.concatenate([frag, _BASES[rng.integers(0, 4, size=max(l1, l2) - len(frag))]])
            if long_indel_frac > 0 and rng.random() < long_indel_frac and len(frag) > 2 * long_indel_max + 40:
                k = int(rng.integers(2, long_indel_max + 1))
                at = int(rng.integers(10, max(l1, l2) - 10)) if rng.random() < 0.5 else len(frag) - int(rng.integers(10, max(l1, l2) - 10))
                at = min(max(at, 5), len(frag) - k - 5)
                if rng.random() < 0.5:
                    frag = np.concatenate([frag[:at], frag[at + k:]])                                   # deletion from the read
                else:
                    frag = np.concatenate([frag[:at], _BASES[rng.integers(0, 4, size=k)], frag[at:]])      # insertion
                if len(frag) < max(l1, l2):
                    frag = np.concatenate([frag, _BASES[rng.integers(0, 4, size=max(l1, l2) - len(frag))]])
            a = frag[:l1]
            b = _COMP[frag[len(frag) - l2:]][::-1]
            if junk_end_frac > 0 and rng.random() < junk_end_frac:
                k = int(rng.integers(3, junk_end_max + 1))
                pick_a = rng.random() < 0.5
                tgt = (a if pick_a else b).copy()
                k = min(k, len(tgt))
                if rng.random() < 0.5:
                    tgt[:k] = _BASES[rng.integers(0, 4, size=k)]
                else:
                    tgt[len(tgt) - k:] = _BASES[rng.integers(0, 4, size=k)]
                if pick_a:
                    a = tgt
                else:
                    b = tgt
            if rng.random() < 0.5:
                a, b = b, a
            if broken_mate_frac > 0 and rng.random() < broken_mate_frac:
                if rng.random() < 0.5:
                    a = _BASES[rng.integers(0, 4, size=l1)]
                else:
                    b = _BASES[rng.integers(0, 4, size=l2)]
            if noisy_mate_frac > 0 and rng.random() < noisy_mate_frac:
                pick_a = rng.random() < 0.5
                tgt = (a if pick_a else b).copy()
                hit = rng.random(len(tgt)) < noisy_sub
                tgt[hit] = _BASES[(np.searchsorted(_BASES, tgt[hit]) + 1 + rng.integers(0, 3, size=int(hit.sum()))) & 3]
                if pick_a:
                    a = tgt
                else:
                    b = tgt
        if n_frac > 0:
            a = a.copy(); b = b.copy()
            a[rng.random(len(a)) < n_frac] = ord("N")
            b[rng.random(len(b)) < n_frac] = ord("N")
        names.append("r%d" % (p + 1))
        r1s.append(a.tobytes())
        r2s.append(b.tobytes())
    return names, r1s, r2s


def write_fastq_pair(path1: str, path2: str, names, r1s, r2s, qual_char: bytes = b"I") -> None:
    with open(path1, "wb") as f1, open(path2, "wb") as f2:
        for n, a, b in zip(names, r1s, r2s):
            f1.write(b"@" + n.encode() + b"/1\n" + a + b"\n+\n" + qual_char * len(a) + b"\n")
            f2.write(b"@" + n.encode() + b"/2\n" + b + b"\n+\n" + qual_char * len(b) + b"\n")
