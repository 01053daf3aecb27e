"""The certificate kernel in front of the single-end rescue filter (kernels_thermo.hip): it may certify an instance only if the exact DP
(callDP, CPU_DP.cpp:122-978, here dp_fill_kernel which the golden vectors pin) stays below the cutoff -- never the other way round."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _comp_rev(codes):
    return (3 - codes)[::-1]


def _make_instances(rng, text, n, kinds, lo=101, hi=160, margin=30):
    """n (read, window) instances of mixed kinds; returns (list of read code arrays as stored, tasks array, kind per instance)"""
    from megapath_amd import hipapi as H
    N = len(text)
    reads, tasks, kind = [], np.zeros(n, dtype=H.DP_TASK_DTYPE), np.zeros(n, dtype=np.int32)
    for t in range(n):
        L = int(rng.integers(lo, hi + 1))
        m = L + 2 * margin
        ws = int(rng.integers(0, N - m - 64))
        win = text[ws:ws + m]
        k = int(kinds[rng.integers(0, len(kinds))])
        rd = rng.integers(0, 4, L).astype(np.uint8)
        if L < 70 and k in (1, 2, 4):
            k = 7           # (the planted shapes below need room: short reads get the plain run of 12 .. 26 instead)
        if k in (0, 7):     # a chance seed: a run of 17 .. 19 (7: 12 .. 26) on the seed's diagonal, nothing else
            r = int(rng.integers(17, 20)) if k == 0 else int(rng.integers(12, 27)); j0 = int(rng.integers(0, L - r + 1))
            rd[j0:j0 + r] = win[margin + j0:margin + j0 + r]
        elif k == 1:        # near the cutoff: the run, then a second exact stretch right after a mismatch or a short gap (same or shifted diagonal)
            r = int(rng.integers(14, 24)); e = int(rng.integers(5, 20)); sh = int(rng.integers(-3, 4))
            j0 = int(rng.integers(0, L - r - e - 8)); j1 = j0 + r + 1 + max(0, -sh)
            rd[j0:j0 + r] = win[margin + j0:margin + j0 + r]
            rd[j1:j1 + e] = win[margin + sh + j1:margin + sh + j1 + e]
        elif k == 2:        # as 1 but the second stretch BEFORE the run
            r = int(rng.integers(14, 24)); e = int(rng.integers(5, 20)); sh = int(rng.integers(-3, 4))
            j0 = int(rng.integers(e + 8, L - r)); j1 = j0 - 1 - e - max(0, sh)
            rd[j0:j0 + r] = win[margin + j0:margin + j0 + r]
            rd[j1:j1 + e] = win[margin + sh + j1:margin + sh + j1 + e]
        elif k == 3:        # a true alignment with a few mismatches and possibly an indel
            rd = win[margin:margin + L].copy()
            for _ in range(int(rng.integers(0, 12))):
                rd[rng.integers(0, L)] = rng.integers(0, 4)
            if rng.random() < 0.5:
                p = int(rng.integers(20, L - 20)); g = int(rng.integers(1, 6))
                rd = np.concatenate([rd[:p], rd[p + g:], rng.integers(0, 4, g).astype(np.uint8)]) if rng.random() < 0.5 else np.concatenate([rd[:p], rng.integers(0, 4, g).astype(np.uint8), rd[p:L - g]])
        elif k == 4:        # stretches on both sides of the run, and a partial alignment of ~ cutoff matches with mismatches inside
            r = int(rng.integers(12, 20)); j0 = int(rng.integers(15, L - r - 15))
            rd[j0:j0 + r] = win[margin + j0:margin + j0 + r]
            a = int(rng.integers(4, 14)); b = int(rng.integers(4, 14))
            rd[j0 - 1 - a:j0 - 1] = win[margin + j0 - 1 - a:margin + j0 - 1]
            rd[j0 + r + 1:j0 + r + 1 + b] = win[margin + j0 + r + 1:margin + j0 + r + 1 + b]
        elif k == 5:        # low complexity: read and window share a short tandem repeat (many diagonals match)
            unit = rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.uint8)
            rep = np.tile(unit, 300)
            a = int(rng.integers(0, min(40, L - 25))); b = min(int(rng.integers(20, 90)), L - a)
            win = win.copy(); win[margin + a:margin + a + b] = rep[:b]
            text[ws:ws + m] = win
            rd[a:a + b] = rep[:b]
        # k == 6: a random read (no run: not eligible)
        strand = 1 + int(rng.integers(0, 2))
        reads.append(rd if strand == 1 else _comp_rev(rd))
        tasks[t] = (t, strand, ws, m, 0)
        kind[t] = k
    return reads, tasks, kind


@pytest.mark.parametrize("kinds,seed,lo,hi", [((0,), 1, 101, 160), ((7, 1, 2, 4), 2, 101, 160), ((0, 1, 2, 3, 4, 5, 6, 7), 3, 101, 160),
                                              ((0,), 4, 161, 320), ((0, 1, 2, 3, 4, 5, 6, 7), 5, 140, 320), ((7, 1, 2, 4), 6, 290, 320),
                                              ((0,), 7, 60, 100), ((0, 1, 2, 3, 4, 5, 6, 7), 8, 45, 100), ((7, 1, 2, 4), 9, 90, 100)])
def test_certificate_never_rejects_an_alignment(kinds, seed, lo, hi):
    from megapath_amd import hipapi as H
    from megapath_amd.index_builder import build_index_arrays
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    text = rng.integers(0, 4, 1 << 20).astype(np.uint8)
    n = (60000 if hi <= 160 else 30000) * int(os.environ.get("MP_THERMO_TEST_SCALE", "1"))      # (one-off campaigns: MP_THERMO_TEST_SCALE=10)
    short = hi <= 100                                                  # round 4: reads of up to 100 bases, windows of read + 2 x 25 (the kernel's margin-25 variant)
    reads, tasks, kind = _make_instances(rng, text, n, kinds, lo, hi, 25 if short else 30)  # (kind 5 plants repeats into the text: build the index afterwards)
    arrs = build_index_arrays([("t", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    seqs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    packed, lens, W = H.pack_reads(seqs, 161 if hi <= 160 else 321)
    ctx.set_reads(packed, lens)
    cert, st = ctx.test_dp_thermo(tasks, 0 if short else -1)
    score = ctx.sgdp_scores(tasks, 130, 130)
    cutoff = np.maximum(30, (0.2 * lens[tasks["read_id"]]).astype(np.int64))
    bad = np.nonzero((cert != 0) & (score >= cutoff))[0]
    assert len(bad) == 0, (bad[:10], score[bad[:10]], kind[bad[:10]])
    assert sum(st) == n and st[0] == int((cert != 0).sum())
    if short:
        # the other variants must refuse these shapes (windows of read + 50): nothing may be certified by the wrong geometry
        c1, st1 = ctx.test_dp_thermo(tasks, 1)
        assert not c1.any() and st1[3] == n
    # the exact scores themselves against the oracle's callDP on a sample (dp_fill_kernel is pinned by the golden vectors; this guards the harness)
    for t in rng.integers(0, n, 40):
        rd = reads[t] if tasks["strand"][t] == 1 else _comp_rev(reads[t])
        ws, m = int(tasks["win_start"][t]), int(tasks["win_len"][t])
        want = O.sgdp(text[ws:ws + m], np.ascontiguousarray(rd), 130, 130)
        got = int(score[t])
        assert (got if got >= cutoff[t] else 0) == (want[0] if want[0] >= cutoff[t] else 0), t
    # usefulness: pure chance seeds are nearly all certified
    if kinds == (0,):
        assert st[0] > (0.9 if hi <= 160 else 0.8) * n, st       # (short reads: smaller tables, the same share and more)
    if 3 in kinds:
        assert not cert[(kind == 3) & (score >= cutoff)].any()
    ctx.close()
    ix.close()


def _make_rescue_instances(rng, text, n, kinds, lo, hi):
    """mate-rescue windows (HalfEndAlgnBatch::pack, DV-DPfunctions.cpp:1151-1231): up to 751 rows, no seed; the mate may be there, partly there, or not"""
    from megapath_amd import hipapi as H
    N = len(text)
    reads, tasks, kind = [], np.zeros(n, dtype=H.DP_TASK_DTYPE), np.zeros(n, dtype=np.int32)
    for t in range(n):
        L = int(rng.integers(lo, hi + 1))
        k = int(kinds[rng.integers(0, len(kinds))])
        m = int(rng.integers(max(L + 1, 300), 752)) if k != 6 else int(rng.integers(1, L + 20))      # 6: a window cut short by the end of the text (shorter than the read, even)
        ws = int(rng.integers(0, N - m - 64))
        win = text[ws:ws + m]
        rd = rng.integers(0, 4, L).astype(np.uint8)
        if k == 1:          # the mate is there: a true alignment somewhere in the window, a few mismatches, possibly an indel, possibly junk ends (soft clips)
            o = int(rng.integers(0, m - L + 1)) if m >= L else 0
            src = win[o:o + L].copy()
            if len(src) == L:
                rd = src
                for _ in range(int(rng.integers(0, 14))):
                    rd[rng.integers(0, L)] = rng.integers(0, 4)
                if rng.random() < 0.5:
                    p = int(rng.integers(10, L - 10)); g = int(rng.integers(1, 6))
                    rd = np.concatenate([rd[:p], rd[p + g:], rng.integers(0, 4, g).astype(np.uint8)]) if rng.random() < 0.5 else np.concatenate([rd[:p], rng.integers(0, 4, g).astype(np.uint8), rd[p:L - g]])
                if rng.random() < 0.3:
                    j = int(rng.integers(5, 40)); rd[:j] = rng.integers(0, 4, j)
        elif k == 2:        # near the cutoff: one exact stretch of 20 .. 34 bases, a second one of 3 .. 14 right after a mismatch or a short gap
            r = int(rng.integers(20, 35)); e = int(rng.integers(3, 15)); sh = int(rng.integers(-3, 4))
            o = int(rng.integers(5, m - L - 5)) if m > L + 12 else 5
            j0 = int(rng.integers(0, L - r - e - 8)); j1 = j0 + r + 1 + max(0, -sh)
            if o + j1 + sh + e < m and o + j0 + r < m:
                rd[j0:j0 + r] = win[o + j0:o + j0 + r]
                rd[j1:j1 + e] = win[o + sh + j1:o + sh + j1 + e]
        elif k == 3:        # a stretch at the very first or the very last rows of the window (the sweep's edges)
            r = int(rng.integers(22, 45))
            if rng.random() < 0.5:
                j0 = int(rng.integers(0, L - r + 1)); rd[j0:j0 + r] = win[:r]
            else:
                j0 = int(rng.integers(0, L - r + 1)); rd[j0:j0 + r] = win[m - r:]
        elif k == 5:        # low complexity: read and window share a short tandem repeat
            unit = rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.uint8)
            rep = np.tile(unit, 400)
            a = int(rng.integers(0, max(1, m - 100))); b = min(int(rng.integers(20, 120)), L, m - a)
            win = win.copy(); win[a:a + b] = rep[:b]
            text[ws:ws + m] = win
            ra = int(rng.integers(0, L - b + 1)); rd[ra:ra + b] = rep[:b]
        # k == 0, 6: a random read: nothing there
        strand = 1 + int(rng.integers(0, 2))
        reads.append(rd if strand == 1 else _comp_rev(rd))
        tasks[t] = (t, strand, ws, m, 0)
        kind[t] = k
    return reads, tasks, kind


@pytest.mark.parametrize("name,lo,hi,margin,variant", [("narrow", 101, 160, 30, 1), ("short", 45, 100, 25, 0), ("wide", 161, 320, 30, 2), ("rescue", 75, 160, 0, 3), ("rescue_wide", 100, 320, 0, 4)])
def test_certificate_is_a_function_of_the_instance_alone(name, lo, hi, margin, variant):
    """The 32 instances of a word share nothing but the number of steps their wavefront sweeps, and more steps add nothing: the same instance must get
    the same answer whatever it is grouped with, and run after run.  (Round 4: the lane-to-lane transfers were one-instruction asm statements whose DPP
    operand could be read within the two wait states after the instruction that wrote it -- a build with another schedule answered a few instances in
    a thousand differently from run to run, in both directions.  Mixed read lengths and a second order of the same instances show that.)"""
    from megapath_amd import hipapi as H
    from megapath_amd.index_builder import build_index_arrays
    rng = np.random.default_rng(99)
    text = rng.integers(0, 4, 1 << 21).astype(np.uint8)
    n = 40000
    if variant >= 3:
        reads, tasks, kind = _make_rescue_instances(rng, text, n, (0, 0, 0, 1, 2, 3, 5, 6), lo, hi)
    else:
        reads, tasks, kind = _make_instances(rng, text, n, (0, 0, 0, 1, 2, 3, 4, 5, 6, 7), lo, hi, margin)
    arrs = build_index_arrays([("t", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    seqs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    packed, lens, W = H.pack_reads(seqs, 161 if hi <= 160 else 321)
    ctx.set_reads(packed, lens)
    tasks = np.ascontiguousarray(tasks)
    first, st = ctx.test_dp_thermo(tasks, variant)
    first = first.copy()
    assert 0.2 * n < st[0] < 0.9 * n, st
    for rep in range(2):
        again, _ = ctx.test_dp_thermo(tasks, variant)
        assert (again == first).all(), (name, rep, int((again != first).sum()))
    # another grouping: sorted by window length (what run_dp_filtered hands over), and a random order
    for order in (np.argsort(tasks["win_len"], kind="stable"), rng.permutation(n)):
        got, _ = ctx.test_dp_thermo(np.ascontiguousarray(tasks[order]), variant)
        back = np.empty(n, dtype=np.uint8)
        back[order] = got
        bad = np.nonzero(back != first)[0]
        assert len(bad) == 0, (name, len(bad), [(int(t), int(first[t]), int(back[t]), int(kind[t]), int(lens[tasks["read_id"][t]]), int(tasks["win_len"][t])) for t in bad[:8]])
    ctx.close()
    ix.close()


@pytest.mark.parametrize("kinds,seed,lo,hi", [((0,), 11, 101, 160), ((0, 1, 2, 3, 5, 6), 12, 75, 160), ((1, 2, 3), 13, 140, 160), ((0,), 14, 161, 320), ((0, 1, 2, 3, 5, 6), 15, 100, 320),
                                              ((1, 2, 3), 16, 250, 320)])
def test_rescue_window_certificate_never_rejects_an_alignment(kinds, seed, lo, hi):
    """round 4: the run-less variants for the windows of the mate rescue (up to 751 rows): certified => the exact DP stays below the cutoff"""
    from megapath_amd import hipapi as H
    from megapath_amd.index_builder import build_index_arrays
    rng = np.random.default_rng(seed)
    text = rng.integers(0, 4, 1 << 20).astype(np.uint8)
    n = (24000 if hi <= 160 else 12000) * int(os.environ.get("MP_THERMO_TEST_SCALE", "1"))
    reads, tasks, kind = _make_rescue_instances(rng, text, n, kinds, lo, hi)
    arrs = build_index_arrays([("t", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    seqs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    packed, lens, W = H.pack_reads(seqs, 161 if hi <= 160 else 321)
    ctx.set_reads(packed, lens)
    cert, st = ctx.test_dp_thermo(tasks, 3 if hi <= 160 else 4)
    score = ctx.sgdp_scores(tasks, 130, 130)
    cutoff = np.maximum(30, (0.2 * lens[tasks["read_id"]]).astype(np.int64))
    bad = np.nonzero((cert != 0) & (score >= cutoff))[0]
    assert len(bad) == 0, (bad[:10], score[bad[:10]], kind[bad[:10]])
    assert sum(st) == n and st[0] == int((cert != 0).sum())
    if kinds == (0,):
        assert st[0] > 0.9 * n, st                      # windows without the mate are what the variant is for
    if 1 in kinds:
        assert (score[kind == 1] >= cutoff[kind == 1]).mean() > 0.8 and not cert[(kind == 1) & (score >= cutoff)].any()
    ctx.close()
    ix.close()


@pytest.mark.parametrize("read_len,len_range", [(151, None), (250, None), (300, (75, 300)), (100, (40, 100))])
def test_rescue_round_equal_with_and_without_certificate(read_len, len_range, monkeypatch):
    """the product path: the same single-end and mate-rescue hits with the certificate on and off (17-base seeds, many random and broken mates)"""
    monkeypatch.setenv("MP_DP_THERMO_RESCUE", "2")        # the mate-rescue variant whatever the batch size and its last trial's yield (run_dp_filtered)
    import ctypes as C
    from megapath_amd import hipapi as H, index_builder, synth
    genome = synth.random_genome(3000000, 77, n_seqs=2, repeat_frac=0.05)
    arrs = index_builder.build_index_arrays(genome, sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    names, r1, r2 = synth.simulate_pairs(genome, 20000 if read_len < 200 else 8000, read_len, 78, sub=0.02, ins=0.001, dele=0.001, random_frac=0.3, broken_mate_frac=0.3,
                                         noisy_mate_frac=0.2, junk_end_frac=0.2, len_range=len_range, frag_mean=read_len + 200.0, frag_max=read_len + 450)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, read_len + 1)
    outs = []
    for on in (1, 0, 2):                              # every variant | off | round 3's coverage (single-end windows of 101 .. 320-base reads)
        ctx = H.Context(ix)
        ctx.set_dp_thermo(on)
        ctx.set_reads(packed, lens)
        ctx.seed_batch(H.MmpParams.nt2())
        ap = H.AlignParams(int(lens.max()), 750, 130, 130, H.MmpParams.nt2())
        ctx.deep_dp_round(ap)
        (sh, ns), (rh, nr), st = ctx.rescue_round(ap)
        a = C.string_at(C.addressof(sh), ns * C.sizeof(H.SingleHit)) if ns else b""
        b = C.string_at(C.addressof(rh), nr * C.sizeof(H.RescueHit)) if nr else b""
        outs.append((a, b, ns, nr, ctx.dp_thermo_stats()))
        ctx.close()
    assert outs[0][2] > 500 and outs[0][3] > 50
    assert outs[0][:4] == outs[1][:4] == outs[2][:4]
    assert outs[0][4][0] > 0 and outs[1][4] == (0, 0, 0, 0)
    assert outs[0][4][0] > outs[2][4][0]              # the mate-rescue windows (and reads of up to 100 bases) are certified too now
    ix.close()
