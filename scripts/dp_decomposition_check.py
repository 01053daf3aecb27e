#!/usr/bin/env python3
"""Randomised check of the decomposition DESIGN.md section 4 uses in its analysis of the rescue filter:
    max(T) <= max( max(T'), max_{a<=b} L(a) + (b - a + 1) + R(b) )
for local alignment with +1/-2, gap -3/-1 (Gotoh), where T is a read x window table that contains an exact match (the seed) on one
diagonal, T' the same table with the seed's cells turned into mismatches, L(a) the best score of a path of T' ending on the cell
diagonally before seed cell a (0 if none), R(b) the best score of a path of T' starting on the cell diagonally after seed cell b.
Vectorised over instances; alphabets of 4 and 2 letters (the latter is tie- and repeat-rich)."""
import sys

import numpy as np

NEG = -10 ** 6


def local_dp(match):
    """match: [N, m, n] bool (window row i, read column j) -> H [N, m+1, n+1] of best local path ENDING at (i, j)."""
    N, m, n = match.shape
    H = np.zeros((N, m + 1, n + 1), dtype=np.int32)
    E = np.full((N, m + 1, n + 1), NEG, dtype=np.int32)      # gap consuming window rows
    F = np.full((N, m + 1, n + 1), NEG, dtype=np.int32)      # gap consuming read columns
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            E[:, i, j] = np.maximum(H[:, i - 1, j] - 3, E[:, i - 1, j] - 1)
            F[:, i, j] = np.maximum(H[:, i, j - 1] - 3, F[:, i, j - 1] - 1)
            d = H[:, i - 1, j - 1] + np.where(match[:, i - 1, j - 1], 1, -2)
            H[:, i, j] = np.maximum(0, np.maximum(d, np.maximum(E[:, i, j], F[:, i, j])))
    return H


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    worst = 0
    for alpha in (4, 2, 3):
        N, m, n = 1500, 56, 40
        win = rng.integers(0, alpha, size=(N, m))
        read = rng.integers(0, alpha, size=(N, n))
        slen = rng.integers(6, 14, size=N)
        q0 = np.array([rng.integers(0, n - s + 1) for s in slen])             # seed start in the read
        w0 = np.array([rng.integers(0, m - s + 1) for s in slen])             # ... and in the window
        for k in range(N):
            read[k, q0[k]:q0[k] + slen[k]] = win[k, w0[k]:w0[k] + slen[k]]
        match = win[:, :, None] == read[:, None, :]
        seed = np.zeros_like(match)
        for k in range(N):
            for t in range(slen[k]):
                seed[k, w0[k] + t, q0[k] + t] = True
        T = local_dp(match).max(axis=(1, 2))
        mp = match & ~seed
        Hf = local_dp(mp)                                                      # paths of T' ending at a cell
        Hb = local_dp(mp[:, ::-1, ::-1])[:, ::-1, ::-1]                        # paths of T' starting at a cell: Hb[:, i, j] = best path starting at cell (i, j) (0-based)
        Tp = Hf.max(axis=(1, 2))
        bound = Tp.copy()
        for k in range(N):
            s = slen[k]
            L = np.array([Hf[k, w0[k] + a, q0[k] + a] for a in range(s)])      # Hf is 1-based: index (w0+a, q0+a) = cell before seed cell a
            R = np.array([Hb[k, w0[k] + b + 1, q0[k] + b + 1] if (w0[k] + b + 1 < m and q0[k] + b + 1 < n) else 0 for b in range(s)])
            best = max(L[a] + (b - a + 1) + R[b] for a in range(s) for b in range(a, s))
            bound[k] = max(bound[k], best)
        viol = int((T > bound).sum())
        slack = int((bound - T).max())
        print("alphabet %d: %d instances, max(T) in %d..%d, max(T') in %d..%d, violations %d, bound exact in %.1f %%, largest slack %d"
              % (alpha, N, T.min(), T.max(), Tp.min(), Tp.max(), viol, 100.0 * (bound == T).mean(), slack))
        worst += viol
    sys.exit(1 if worst else 0)


if __name__ == "__main__":
    main()
