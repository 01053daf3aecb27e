#!/usr/bin/env python3
"""Prototype (numpy) of an exact cheaper rejection for the single-end rescue filter, used to measure how often it certifies before any
kernel is written.  Instance: read (L) x window (m = L + 60) with an exact run (the seed) on diagonal 30.
  T'  = the table with the seed run's cells turned into mismatches;  H' = local Gotoh scores of T' (floor at 0 everywhere: over-estimate)
  bound = max( max H',  max_a (H'(cell before run cell a) - a) + len + max{H'(i,j): i > i0, j > j0} )       (i0, j0 = first run cell)
and the instance is rejected when bound < cutoff (30).  Checks bound >= true max of T on every instance and reports the certification rate."""
import sys

import numpy as np

NEG = -10 ** 6


def local_dp(match):
    N, m, n = match.shape
    H = np.zeros((N, m + 1, n + 1), dtype=np.int16)
    E = np.full((N, n + 1), NEG, dtype=np.int32)
    for i in range(1, m + 1):
        F = np.full(N, NEG, dtype=np.int32)
        Hrow_prev = H[:, i - 1, :].astype(np.int32)
        Hrow = np.zeros((N, n + 1), dtype=np.int32)
        for j in range(1, n + 1):
            E[:, j] = np.maximum(Hrow_prev[:, j] - 3, E[:, j] - 1)
            F = np.maximum(Hrow[:, j - 1] - 3, F - 1)
            d = Hrow_prev[:, j - 1] + np.where(match[:, i - 1, j - 1], 1, -2)
            Hrow[:, j] = np.maximum(0, np.maximum(d, np.maximum(E[:, j], F)))
        H[:, i, :] = Hrow
    return H


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    N, L = 3000, 150
    m = L + 60
    win = rng.integers(0, 4, size=(N, m))
    read = rng.integers(0, 4, size=(N, L))
    # chance seeds: length 17 + geometric(3/4), anywhere in the read, on diagonal 30 (window row = read column + 30)
    slen = 17 + rng.geometric(0.75, size=N) - 1
    slen = np.minimum(slen, 26)
    q0 = np.array([rng.integers(0, L - s + 1) for s in slen])
    for k in range(N):
        read[k, q0[k]:q0[k] + slen[k]] = win[k, q0[k] + 30:q0[k] + 30 + slen[k]]
        # maximal: the flanking cells on the diagonal mismatch
        if q0[k] > 0 and read[k, q0[k] - 1] == win[k, q0[k] + 29]:
            read[k, q0[k] - 1] = (read[k, q0[k] - 1] + 1) % 4
        e = q0[k] + slen[k]
        if e < L and read[k, e] == win[k, e + 30]:
            read[k, e] = (read[k, e] + 1) % 4
    match = win[:, :, None] == read[:, None, :]
    T = local_dp(match).max(axis=(1, 2)).astype(int)
    mp = match.copy()
    for k in range(N):
        for t in range(slen[k]):
            mp[k, q0[k] + 30 + t, q0[k] + t] = False
    Hp = local_dp(mp).astype(int)
    Tp = Hp.max(axis=(1, 2))
    bound = np.zeros(N, dtype=int)
    for k in range(N):
        i0, j0, s = q0[k] + 30, q0[k], slen[k]                          # 0-based first run cell (row i0, col j0); Hp is 1-based
        Lm = max(Hp[k, i0 + a, j0 + a] - a for a in range(s))           # Hp[i0 + a, j0 + a] = cell before run cell a
        Rs = Hp[k, i0 + 2:, j0 + 2:].max() if (i0 + 2 <= m and j0 + 2 <= L) else 0     # rows > i0, cols > j0 (0-based) = 1-based >= i0+2
        bound[k] = max(Tp[k], Lm + s + Rs)
    viol = int((T > bound).sum())
    cert = bound < 30
    print("instances %d: seed lengths %s" % (N, np.bincount(slen)[17:].tolist()))
    print("true max(T): mean %.1f, >= 30: %d;  max(T'): mean %.1f, max %d" % (T.mean(), int((T >= 30).sum()), Tp.mean(), Tp.max()))
    print("violations (true > bound): %d;  certified (bound < 30): %.1f %%;  mean slack bound - true: %.1f" % (viol, 100.0 * cert.mean(), (bound - T).mean()))
    print("max(T') histogram from 4:", np.bincount(Tp)[4:].tolist(), " bound histogram from 20:", np.bincount(bound)[20:].tolist())
    for s in range(17, 23):
        sel = slen == s
        if sel.any():
            print("  seed %d: %d instances, certified %.1f %%" % (s, int(sel.sum()), 100.0 * cert[sel].mean()))
    sys.exit(1 if viol else 0)


if __name__ == "__main__":
    main()
