"""The no-table short cuts of the deep-DP pre-filter (kernels_dp_band.hip::dp_prefilter_kernel: one mismatch on the only matching diagonal =>
score L - 3 in one cell, two mismatches well inside the read => L - 6) rest on a hand case analysis over single-gap and two-event alternatives.
The soak and test_one_mismatch_instances_need_no_table sample those cases; this test ENUMERATES them around the proof's boundaries (advisor
finding of round 3): every mismatch column, periodic stretches of every critical length around it (shifted diagonals then match as well:
the single-gap alternatives), one-base insertions / deletions inside such stretches (true ties with the mismatch alignment), grids of
mismatch pairs -- each instance through the band scheme (mp_test_dp_band) and through the full-table kernels the golden callDP vectors pin
(mp_sgdp_batch): score, start offset and tie count must be equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LETTERS = np.arange(4, dtype=np.uint8)


def _comp_rev(codes):
    return (3 - codes)[::-1]


def _periodic(rng, n, p):
    unit = rng.integers(0, 4, p).astype(np.uint8)
    if p > 1 and (unit == unit[0]).all():
        unit[-1] = (unit[0] + 1) & 3
    return np.resize(unit, n)


def _other(base, k=1):
    return np.uint8((int(base) + k) & 3)


def _instances(rng, L):
    """(read, window) pairs: the read lies at window offset 30 (the deep-DP windows are read + 2 x 30) apart from the planted differences"""
    b = 9 + (L - 1) // 32 if L > 64 else 9                     # about the block size the pre-filter uses; the critical stretch lengths are taken around 2b
    out = []

    def base():
        r = rng.integers(0, 4, L).astype(np.uint8)
        w = np.concatenate([rng.integers(0, 4, 30).astype(np.uint8), r, rng.integers(0, 4, 30).astype(np.uint8)])
        return r, w
    # F1: exactly one mismatch, at every column
    r0, w0 = base()
    for c in range(L):
        w = w0.copy(); w[30 + c] = _other(w[30 + c], 1 + c % 3)
        out.append((r0, w))
    crit = sorted({2, 3, 5, b - 1, b, b + 1, 2 * b - 3, 2 * b - 2, 2 * b - 1, 2 * b, 2 * b + 1, 2 * b + 4})
    # F2: one mismatch inside / at the edge of / next to a periodic stretch (periods 1 .. 3) of every critical length
    for p in (1, 2, 3):
        for e in crit:
            for c in list(range(2, L - 2, max(1, L // 12))) + [3, 4, L - 4, L - 3, L - 2 * b + 1, L - 2 * b + 2, L - 2 * b + 3, 2 * b - 2, 2 * b - 1, 2 * b]:
                if c < 0 or c >= L:
                    continue
                for where in (0, 1, 2):                      # the stretch starts at the mismatch, ends at it, is centred on it
                    a = c if where == 0 else c - e + 1 if where == 1 else c - e // 2
                    a = max(0, min(a, L - e))
                    r, w = base()
                    pat = _periodic(rng, e, p)
                    r = r.copy(); r[a:a + e] = pat
                    w = w.copy(); w[30:30 + L] = r
                    w[30 + c] = _other(w[30 + c], 1 + (c + e) % 3)
                    out.append((r, w))
    # F3: a one-base insertion or deletion of the WINDOW inside a periodic stretch (the gapped alignment and a mismatch alignment tie or nearly tie),
    # with and without a further mismatch elsewhere
    for p in (1, 2):
        for e in (4, b, 2 * b - 1, 2 * b + 2):
            for a in range(3, L - e - 3, max(1, L // 9)):
                for extra in (None, 5, L - 6, a + e + 7):
                    r, w = base()
                    r = r.copy(); r[a:a + e] = _periodic(rng, e, p)
                    body = r.copy()
                    j = a + e // 2
                    for ins in (True, False):
                        wb = np.concatenate([body[:j], body[j:j + 1], body[j:]]) if ins else np.concatenate([body[:j], body[j + 1:]])
                        wb = wb[:L] if len(wb) >= L else np.concatenate([wb, rng.integers(0, 4, L - len(wb)).astype(np.uint8)])
                        ww = np.concatenate([w[:30], wb, w[30 + L:]])
                        if extra is not None and 0 <= extra < L:
                            ww[30 + extra] = _other(ww[30 + extra])
                        out.append((r, ww))
    # F4: two mismatches on a grid of column pairs (the L - 6 short cut wants both well inside the read), plain and with a periodic stretch between them
    cols = sorted({1, 3, 2 * b - 2, 2 * b - 1, 2 * b, 4 * b - 3, 4 * b - 2, 4 * b - 1, L // 2, L - 4 * b + 1, L - 4 * b + 2, L - 2 * b, L - 2 * b + 1, L - 2 * b + 2, L - 4, L - 2} | set(range(5, L - 5, max(1, L // 10))))
    cols = [c for c in cols if 0 <= c < L]
    for i, c1 in enumerate(cols):
        for c2 in cols[i + 1:]:
            r, w = base()
            if (c1 + c2) % 3 == 0 and c2 - c1 > 6:
                r = r.copy(); r[c1 + 1:c2] = _periodic(rng, c2 - c1 - 1, 1 + (c1 % 3)); w = w.copy(); w[30:30 + L] = r
            w = w.copy(); w[30 + c1] = _other(w[30 + c1]); w[30 + c2] = _other(w[30 + c2], 2)
            out.append((r, w))
    return out


@pytest.mark.parametrize("L,seed", [(149, 1), (150, 2), (100, 3), (61, 4), (250, 5)])
def test_no_table_short_cuts_equal_full_tables_on_enumerated_alternatives(L, seed):
    from megapath_amd import hipapi as H
    from megapath_amd.index_builder import build_index_arrays
    rng = np.random.default_rng(seed)
    inst = _instances(rng, L)
    n = len(inst)
    m = L + 60
    text = np.concatenate([w for _, w in inst] + [rng.integers(0, 4, 4096).astype(np.uint8)])
    arrs = build_index_arrays([("t", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    tasks = np.zeros(n, dtype=H.DP_TASK_DTYPE)
    reads = []
    for t, (r, _) in enumerate(inst):
        strand = 1 + (t & 1)
        reads.append(r if strand == 1 else _comp_rev(r))
        tasks[t] = (t, strand, t * m, m, 0)
    seqs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    packed, lens, W = H.pack_reads(seqs, L + 1)
    ctx.set_reads(packed, lens)
    full = ctx.sgdp_results(tasks, 130, 130, band=False)
    for one_mm in ("2", "0"):                                    # the short cuts on (default) and off (every such instance takes the first band pass): both equal the full tables
        import os
        os.environ["MP_DP_ONE_MISMATCH"] = one_mm
        try:
            band = ctx.sgdp_results(tasks, 130, 130, band=True)
        finally:
            os.environ.pop("MP_DP_ONE_MISMATCH", None)
        bad = np.nonzero((band["score"] != full["score"]) | ((full["score"] > 0) & ((band["hit_loc"] != full["hit_loc"]) | (band["max_score_count"] != full["max_score_count"]))))[0]
        assert len(bad) == 0, (one_mm, L, len(bad), [(int(t), tuple(int(x) for x in band[t])[:3], tuple(int(x) for x in full[t])[:3]) for t in bad[:8]])
    # the enumeration does reach the short cuts: many instances score exactly L - 3 and L - 6
    assert (full["score"] == L - 3).sum() > L // 2 and (full["score"] == L - 6).sum() > 20
    ctx.close()
    ix.close()
