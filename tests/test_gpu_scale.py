"""-m gpu: parity at larger, nastier sizes than the committed goldens, and size-independent properties at batch scale.

* whole-path stdout of the product == the oracle's, on freshly generated genomes with copied repeats, indels, N bases,
  random/broken mates and mixed read lengths (exercises slab overflow, heap-sort merges, the wide DP templates, rescue);
* at 2^18 pairs against a 64 Mbp genome (no oracle at this size): error-free pairs must come back as a hit with score
  L+L at exactly the simulated coordinates; splitting the batch must not change any pair's hit set; two runs are identical.
"""
import ctypes as C
import io
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import fastq_records
from megapath_amd import hipapi as H
from megapath_amd import index_builder, pipeline, synth

pytestmark = pytest.mark.gpu


def _build(tmp_path, genome):
    arrs = index_builder.build_index_arrays(genome, sa_interval=16)
    prefix = str(tmp_path / "ref.index")
    index_builder.write_index_files(prefix, arrs)
    return prefix


@pytest.mark.parametrize("cfg", [
    dict(tag="div", n_bases=1200000, n_seqs=5, n_pairs=4000, read_len=150, L=151, sim=dict(sub=0.03, ins=0.003, dele=0.003, random_frac=0.05,
                                                                                        broken_mate_frac=0.15, n_frac=0.002)),
    dict(tag="mixed", n_bases=800000, n_seqs=3, n_pairs=2000, read_len=150, L=301, sim=dict(sub=0.02, ins=0.002, dele=0.002, random_frac=0.05,
                                                                                         broken_mate_frac=0.3, len_range=(75, 300), frag_mean=450,
                                                                                         frag_sd=60, frag_max=700)),
    dict(tag="noisy", n_bases=600000, n_seqs=40, n_pairs=3000, read_len=150, L=151, sim=dict(sub=0.02, ins=0.002, dele=0.002, random_frac=0.03,
                                                                                          broken_mate_frac=0.05, noisy_mate_frac=0.4, noisy_sub=0.2,
                                                                                          n_frac=0.001)),
    # long indels and junk ends: alignments far off the window's main diagonal -- the banded DP kernel must hand exactly
    # the right instances to the full-table kernels; mixed lengths <= 150 exercise both DP margins (25 / 30)
    dict(tag="gappy", n_bases=700000, n_seqs=4, n_pairs=4000, read_len=150, L=151, sim=dict(sub=0.01, ins=0.002, dele=0.002, random_frac=0.02,
                                                                                         broken_mate_frac=0.05, long_indel_frac=0.5, long_indel_max=26,
                                                                                         junk_end_frac=0.3, junk_end_max=45, len_range=(90, 150))),
    dict(tag="tandem", n_bases=400000, n_seqs=3, n_pairs=3000, read_len=150, L=151, tandem=500,
         sim=dict(sub=0.01, ins=0.002, dele=0.002, random_frac=0.02, broken_mate_frac=0.05, long_indel_frac=0.2, long_indel_max=15)),
])
def test_whole_path_equals_oracle_on_fresh_data(tmp_path, cfg, monkeypatch):
    monkeypatch.setenv("MP_KMER_FILTER", "1")          # the in-process run uses the 17-mer pre-filter and the denser SA
    monkeypatch.setenv("MP_LKT16", "1")                # ... and the 16-mer look-up table behind the pre-filter
    monkeypatch.setenv("MP_KMER15", "1")               # ... and the per-15-mer arrangement of the pre-filter bits
    monkeypatch.setenv("MP_SA_DENSE", "1")             # samples, the C++ run below neither
    genome = synth.random_genome(cfg["n_bases"], 501, n_seqs=cfg["n_seqs"], repeat_frac=0.12)
    if cfg.get("tandem"):
        # tandem repeats: several copies of a 7..45-base unit inside one DP window (ties among equal alignments: first
        # maximum by row then column, traceback priorities, num_sameScore in the BAM records)
        rng = np.random.Generator(np.random.PCG64(77))
        g2 = []
        for name, arr in genome:
            arr = arr.copy()
            for _ in range(cfg["tandem"]):
                u = int(rng.integers(7, 46)); k = int(rng.integers(2, 6)); p = int(rng.integers(0, len(arr) - u * (k + 1)))
                for c in range(1, k + 1):
                    arr[p + c * u:p + (c + 1) * u] = arr[p:p + u]
            g2.append((name, arr))
        genome = g2
    prefix = _build(tmp_path, genome)
    names, r1, r2 = synth.simulate_pairs(genome, cfg["n_pairs"], cfg["read_len"], 502, **cfg["sim"])
    f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    # oracle
    lib = O.oracle()
    oix = lib.mpo_index_open(prefix.encode())
    p = O.default_params(L=cfg["L"])
    st = O.AlignStats()
    want_path = str(tmp_path / "oracle.fq")
    assert lib.mpo_align_fastq_bam(oix, C.byref(p), f1.encode(), f2.encode(), want_path.encode(), None, str(tmp_path / "obam").encode(), 1,
                                   b"rg1", C.byref(st)) == 0
    lib.mpo_index_close(oix)
    assert st.n_deep_dp_pairs > 0 and st.n_default_dp_pairs > 0 and st.n_unpaired_pairs > 0
    # product
    gix = H.Index(prefix, 0)
    params = pipeline.Soap4Params(max_read_length=cfg["L"], insert_high=750)
    buf = io.BytesIO()
    pst = pipeline.align_fastq(gix, params, f1, f2, buf, batch_pairs=1500)       # several batches on purpose
    gix.close()
    got_path = tmp_path / "got.fq"
    got_path.write_bytes(buf.getvalue())
    got, want = fastq_records(str(got_path)), fastq_records(want_path)
    assert len(got) == len(want) == 2 * cfg["n_pairs"]
    assert got == want
    assert pst.n_deep_pairs == st.n_deep_dp_pairs
    # the C++ driver (bin/soap4-mi355x) on the same inputs, again in several batches
    import subprocess
    from conftest import ROOT
    cxx_out = tmp_path / "cxx.fq"
    env = dict(os.environ, MP_BATCH_PAIRS="1300", MP_KMER_FILTER="0", MP_SA_DENSE="0", MP_LKT16="0", MP_KMER15="0")
    env.pop("MP_SOAP4_PYTHON", None)
    with open(cxx_out, "wb") as fo:
        r = subprocess.run([os.path.join(ROOT, "bin", "soap4"), "pair", prefix, f1, f2, "-o", str(tmp_path / "o"), "-C",
                            os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(cfg["L"]), "-T", "1", "-u", "750", "-F", "-nc",
                            "-b", "-p", "-D", "rg1", "-A", "default"],
                           stdout=fo, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert fastq_records(str(cxx_out)) == want
    # BAM side files of the same run (row a15) against the oracle's records, NM/MD included; the batches differ, so by content
    import bamtext
    n_rec = 0
    for kind, path in (("dpout", "o.dpout.1"), ("gout", "o.gout.1"), ("unpair", "o.unpair")):
        got_recs = bamtext.read_bam(str(tmp_path / path))[2]
        want_recs = open(tmp_path / ("obam.%s.txt" % kind)).read().splitlines()
        assert sorted(got_recs) == sorted(want_recs), kind
        n_rec += len(got_recs)
        if cfg["tag"] == "noisy":
            assert len(got_recs) > 100, kind
    assert n_rec == 2 * cfg["n_pairs"]


def test_batch_scale_properties():
    import torch
    from megapath_amd import simulate
    dev = torch.device("cuda", 0)
    n_bases, P, L = 64_000_000, 1 << 18, 150
    text = simulate.genome(n_bases, seed=11, device=dev, repeat_frac=0.0)
    arrs = index_builder.build_index_arrays([("g", text.cpu().numpy())], sa_interval=16, device=dev)
    gix = H.Index.from_arrays(0, **arrs)
    # error-free FR pairs at known coordinates
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    pos = (torch.rand(P, device=dev, generator=g) * (n_bases - 1000)).to(torch.int64)
    flen = torch.randint(250, 500, (P,), device=dev, generator=g)
    i = torch.arange(L, device=dev).unsqueeze(0)
    left = text[pos.unsqueeze(1) + i]
    right = (3 - text[(pos + flen - L).unsqueeze(1) + i]).flip(1)
    codes = torch.stack([left, right], dim=1).reshape(2 * P, L).contiguous()
    W = (L + 1 + 15) // 16
    packed = simulate.pack_codes(codes, W)
    lens = torch.full((2 * P,), L, dtype=torch.int32, device=dev)
    ap = H.AlignParams(L, 750, 130, 130, H.MmpParams.nt2(), 0, 0)

    def run(lo, hi):
        ctx = H.Context(gix)
        sub_p, sub_l = packed[2 * lo:2 * hi].contiguous(), lens[2 * lo:2 * hi].contiguous()
        ctx.set_reads_device(sub_p.data_ptr(), W, sub_l.data_ptr(), 2 * (hi - lo), L, keep=(sub_p, sub_l))
        hits, nh, st = ctx.deep_dp_round(ap)
        a = np.frombuffer(hits, dtype=np.dtype([("pair", "<u4"), ("lm", "<u4"), ("a1", "<u8"), ("a2", "<u8"), ("s1", "<i4"), ("s2", "<i4"),
                                                ("x", "V16")]), count=nh).copy()
        ctx.close()
        a["pair"] += lo
        key = np.stack([a["pair"].astype(np.int64), a["a1"].astype(np.int64), a["a2"].astype(np.int64), a["s1"].astype(np.int64),
                        a["s2"].astype(np.int64)], axis=1)
        return np.unique(key, axis=0), st

    full, st = run(0, P)
    # 1. every error-free pair is found with full score at the simulated coordinates
    want = np.stack([np.arange(P), pos.cpu().numpy(), (pos + flen - L).cpu().numpy(), np.full(P, L), np.full(P, L)], axis=1)
    have = set(map(tuple, full.tolist()))
    missing = [tuple(w) for w in want.tolist() if tuple(w) not in have]
    assert not missing, missing[:5]
    # 2. run-to-run determinism
    again, _ = run(0, P)
    assert np.array_equal(full, again)
    # 3. batch split invariance: pairs are independent units -- the shard boundary cannot matter
    a, _ = run(0, P // 3)
    b, _ = run(P // 3, P)
    assert np.array_equal(full, np.concatenate([a, b]))
    assert st.seed.n_occ > 0 and st.n_candidates >= P
    gix.close()


@pytest.mark.parametrize("clips,banded,read_len", [((130, 130), True, 150), ((40, 13), True, 150), ((13, 60), True, 150), ((12, 130), False, 150),
                                                  ((130, 130), True, 250), ((130, 130), True, 300), ((130, 130), False, 340),
                                                  ((130, 130), True, 80)])      # reads too short for the block pre-filter: first pass over the whole band
def test_banded_dp_kernel_changes_nothing(tmp_path, clips, banded, read_len):
    """The deep-DP legs with the banded fill kernel (plus full-table fallback) against the full-table kernels alone, on
    reads with long indels and junk ends: the same hit set (positions, scores, windows) in every field, for several
    soft-clip limits -- below MP_BAND_GB, or with reads longer than MP_BAND_MAXL (320), the band path must not engage at all."""
    genome = synth.random_genome(500000, 611, n_seqs=3, repeat_frac=0.1)
    arrs = index_builder.build_index_arrays(genome, sa_interval=16)
    gix = H.Index.from_arrays(0, **arrs)
    names, r1, r2 = synth.simulate_pairs(genome, 3000, read_len, 612, sub=0.01, ins=0.002, dele=0.002, random_frac=0.02, long_indel_frac=0.5,
                                         long_indel_max=26, junk_end_frac=0.3, junk_end_max=45, len_range=(read_len * 3 // 5, read_len),
                                         frag_mean=read_len + 200, frag_max=read_len + 450)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, read_len + 1)

    def run(band):
        ctx = H.Context(gix)
        ctx.set_reads(packed, lens)
        ctx.set_dp_band(band)
        ap = H.AlignParams(1, 750, clips[0], clips[1], H.MmpParams.nt2())
        hits, nh, st = ctx.deep_dp_round(ap)
        got = sorted((h.pair_id, h.left_is_mate, h.algnmt_1, h.algnmt_2, h.score_1, h.score_2, h.start_1, h.start_2, h.dplen_1, h.dplen_2)
                     for h in hits[:nh])
        stats = ctx.dp_band_stats()
        ctx.close()
        return got, stats, st
    full, s_full, st_full = run(False)
    band, s_band, st_band = run(True)
    assert s_full == (0, 0)
    assert len(full) > 1500
    assert band == full
    assert st_band.n_candidates == st_full.n_candidates
    if banded:
        assert s_band[0] == 2 * st_band.n_candidates      # both legs of every candidate went through the band kernel
        assert 0 < s_band[1] < s_band[0] // 2            # some, not most, instances needed the full table
    else:
        assert s_band == (0, 0)
    gix.close()


@pytest.mark.parametrize("sub,junk", [(0.04, 0.0), (0.02, 0.5)])
def test_plain_diagonal_alignments_need_no_traceback(monkeypatch, sub, junk):
    """Deep-DP instances whose reported alignment is the best cell's diagonal (mismatches only, no clip on the left) are finished by
    dp_band_classify_kernel from the read and the window alone (mp_diag_alignment).  Many substitutions -- also within the first bases,
    where the prefix condition fails -- and junk read starts: the hit set must equal what the second pass + traceback give
    (MP_DP_DIAG=0) and what the full-table kernels give, in every field."""
    genome = synth.random_genome(400000, 911, n_seqs=2, repeat_frac=0.1)
    arrs = index_builder.build_index_arrays(genome, sa_interval=16)
    gix = H.Index.from_arrays(0, **arrs)
    names, r1, r2 = synth.simulate_pairs(genome, 4000, 150, 912, sub=sub, ins=0.0005, dele=0.0005, random_frac=0.02, junk_end_frac=junk, junk_end_max=40)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, 151)

    def run(band, diag):
        monkeypatch.setenv("MP_DP_DIAG", "1" if diag else "0")
        ctx = H.Context(gix)
        ctx.set_reads(packed, lens)
        ctx.set_dp_band(band)
        hits, nh, st = ctx.deep_dp_round(H.AlignParams(1, 750, 130, 130, H.MmpParams.nt2()))
        got = sorted((h.pair_id, h.left_is_mate, h.algnmt_1, h.algnmt_2, h.score_1, h.score_2, h.start_1, h.start_2, h.dplen_1, h.dplen_2, h.count_1, h.count_2)
                     for h in hits[:nh])
        ctx.close()
        return got, st
    full, _ = run(False, True)
    two_pass, _ = run(True, False)
    short, _ = run(True, True)
    assert len(full) > 3000
    assert [h[:10] for h in two_pass] == [h[:10] for h in full]      # (the tie counts are only filled in when asked for: not by the full-table kernels here)
    assert short == two_pass                                           # every field, tie counts included
    gix.close()


@pytest.mark.parametrize("low_complexity", [False, True])
def test_one_mismatch_instances_need_no_table(monkeypatch, low_complexity):
    """A deep-DP instance whose read lies on ONE matching diagonal of its window with exactly one mismatch, far enough from both read ends, is
    finished by dp_prefilter_kernel without any table (score L - 3 in the single cell at the diagonal's end; two mismatches well inside the read: L - 6; kernels_dp_band.hip has the
    arguments).  Reads with about one substitution each, on a plain genome and on one riddled with homopolymer runs and two-base tandem
    repeats (shifted diagonals match there, the case the conditions have to keep out): the hit set equals the first pass's
    (MP_DP_ONE_MISMATCH=0) in every field, tie counts included, and the full-table kernels'."""
    genome = synth.random_genome(400000, 1311, n_seqs=2, repeat_frac=0.1)
    if low_complexity:
        rng = np.random.Generator(np.random.PCG64(1312))
        letters = np.frombuffer(b"ACGT", dtype=np.uint8)
        for _, arr in genome:
            for at in range(40, len(arr) - 60, 90):
                at += int(rng.integers(0, 50))
                ln = int(rng.integers(4, 30))
                if rng.random() < 0.5:
                    arr[at:at + ln] = letters[rng.integers(0, 4)]
                else:
                    arr[at:at + ln] = np.resize(letters[rng.integers(0, 4, size=int(rng.integers(2, 4)))], ln)
    arrs = index_builder.build_index_arrays(genome, sa_interval=16)
    gix = H.Index.from_arrays(0, **arrs)
    names, r1, r2 = synth.simulate_pairs(genome, 5000, 150, 1313, sub=0.007, ins=0.0004, dele=0.0004, random_frac=0.02)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, 151)

    def run(band, one_mm):
        monkeypatch.setenv("MP_DP_ONE_MISMATCH", str(one_mm))
        ctx = H.Context(gix)
        ctx.set_reads(packed, lens)
        ctx.set_dp_band(band)
        hits, nh, st = ctx.deep_dp_round(H.AlignParams(1, 750, 130, 130, H.MmpParams.nt2()))
        got = sorted((h.pair_id, h.left_is_mate, h.algnmt_1, h.algnmt_2, h.score_1, h.score_2, h.start_1, h.start_2, h.dplen_1, h.dplen_2, h.count_1, h.count_2)
                     for h in hits[:nh])
        ctx.close()
        return got
    full = run(False, 2)
    first_pass = run(True, 0)
    one = run(True, 1)                                                        # one-mismatch instances without a table
    no_table = run(True, 2)                                                   # and two mismatches well inside the read (the default)
    assert len(full) > 4000
    assert sum(1 for h in no_table if 147 in (h[4], h[5])) > 1000 and sum(1 for h in no_table if 144 in (h[4], h[5])) > 500   # scores L - 3, L - 6
    assert [h[:10] for h in first_pass] == [h[:10] for h in full]
    assert one == first_pass and no_table == first_pass
    gix.close()


def test_seed_projection_wrapping_below_zero(tmp_path):
    """Reads whose first bases are junk and whose match starts within the first bases of the text: the seed projection
    position - query offset wraps below zero in unsigned arithmetic, exactly as in the reference
    (DV-DPfunctions.cpp:2489).  The device keeps positions in 32 bits and restores such values; the whole path must still
    print what the oracle prints."""
    genome = synth.random_genome(300000, 701, n_seqs=1)
    prefix = _build(tmp_path, genome)
    text = genome[0][1]
    rng = np.random.Generator(np.random.PCG64(702))
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    names, r1, r2 = [], [], []
    for p in range(120):
        junk = int(rng.integers(5, 60))
        start = int(rng.integers(0, 12))                       # match starts at text position 0..11
        a = np.concatenate([bases[rng.integers(0, 4, size=junk)], text[start:start + 150 - junk]])
        fl = int(rng.integers(300, 500))
        b = np.asarray([comp[int(x)] for x in text[start + fl - 150:start + fl][::-1]], dtype=np.uint8)
        if p % 2:
            a, b = b, a
        names.append("w%d" % (p + 1)); r1.append(a.tobytes()); r2.append(b.tobytes())
    f1, f2 = str(tmp_path / "w_1.fq"), str(tmp_path / "w_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    lib = O.oracle()
    oix = lib.mpo_index_open(prefix.encode())
    p = O.default_params(L=151)
    st = O.AlignStats()
    want_path = str(tmp_path / "oracle.fq")
    assert lib.mpo_align_fastq(oix, C.byref(p), f1.encode(), f2.encode(), want_path.encode(), None, C.byref(st)) == 0
    lib.mpo_index_close(oix)
    gix = H.Index(prefix, 0)
    # the wrapped projections must really occur in this batch
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, 151)
    ctx = H.Context(gix)
    ctx.set_reads(packed, lens)
    ctx.seed_batch(H.MmpParams.nt2())
    sp, off = ctx.get_seeds()
    assert sum(1 for i in range(int(off[-1])) if sp[i].pos >= (1 << 63)) > 20
    ctx.close()
    buf = io.BytesIO()
    pipeline.align_fastq(gix, pipeline.Soap4Params(max_read_length=151, insert_high=750), f1, f2, buf, batch_pairs=500)
    gix.close()
    got_path = tmp_path / "got.fq"
    got_path.write_bytes(buf.getvalue())
    assert fastq_records(str(got_path)) == fastq_records(want_path)


@pytest.mark.parametrize("rnd", [0, 1, 2, 3])
def test_banded_dp_on_tandem_repeats(rnd, monkeypatch):
    """Genomes with tandem repeats (several copies of a unit inside one DP window: the first-maximum rule has to pick among
    equal alignments, the block pre-filter sees matches on several diagonals), random error profiles and read lengths:
    band kernels vs full-table kernels, without and with tie counts (scripts/band_stress.py runs many more rounds)."""
    monkeypatch.setenv("MP_KMER_FILTER", "1")
    monkeypatch.setenv("MP_LKT16", "1")
    rng = np.random.Generator(np.random.PCG64(9000 + rnd))
    genome = []
    for name, arr in synth.random_genome(300000, 100 + rnd, n_seqs=3, repeat_frac=0.15):
        arr = arr.copy()
        for _ in range(300):
            u = int(rng.integers(7, 46)); k = int(rng.integers(2, 6)); p = int(rng.integers(0, len(arr) - u * (k + 1)))
            for c in range(1, k + 1):
                arr[p + c * u:p + (c + 1) * u] = arr[p:p + u]
        genome.append((name, arr))
    read_len = [150, 100, 200, 250][rnd]
    sim = dict(sub=[0.01, 0.04, 0.003, 0.02][rnd], ins=0.002, dele=0.002, random_frac=0.02, broken_mate_frac=0.05, long_indel_frac=[0.2, 0.0, 0.5, 0.2][rnd],
               long_indel_max=[15, 6, 26, 15][rnd], junk_end_frac=0.3, junk_end_max=45, len_range=(read_len * 3 // 5, read_len), frag_mean=read_len + 200,
               frag_max=read_len + 450)
    names, r1, r2 = synth.simulate_pairs(genome, 2500, read_len, 500 + rnd, **sim)
    gix = H.Index.from_arrays(0, **index_builder.build_index_arrays(genome, sa_interval=16))
    packed, lens, W = H.pack_reads([x for pr in zip(r1, r2) for x in pr], read_len + 1)
    res = {}
    for band in (False, True):
        for counts in (0, 1):
            ctx = H.Context(gix)
            ctx.set_reads(packed, lens)
            ctx.set_dp_band(band)
            hits, nh, st = ctx.deep_dp_round(H.AlignParams(1, 750, 130, 130, H.MmpParams.nt2(), counts, 0))
            rows = sorted((h.pair_id, h.left_is_mate, h.algnmt_1, h.algnmt_2, h.score_1, h.score_2, h.start_1, h.start_2, h.dplen_1, h.dplen_2,
                           h.count_1, h.count_2) for h in hits[:nh])
            res[(band, counts)] = rows if counts else [r[:10] for r in rows]     # the text-mode full-table kernel produces no tie counts
            ctx.close()
    gix.close()
    assert len(res[(True, 1)]) > 1000
    assert res[(True, 0)] == res[(False, 0)]
    assert res[(True, 1)] == res[(False, 1)]


def test_parity_properties_at_human_genome_scale():
    """A 2.3 Gbp synthetic genome -- positions beyond 2^31, every optional table at its real size (2 GiB bitmap, 8.6 GB per-15-mer
    bits, 17 GB 16-mer table, the whole suffix array).  No oracle can run here; what must hold regardless of size:
    the 16-mer table equals 13-mer table + three steps for the 16-mer at EVERY text position; error-free pairs come back at
    exactly their simulated coordinates with score = read length on both mates, also beyond 2^31; the hits are the same with
    every optional table ignored (the reference's own walk); random pairs find nothing."""
    import torch
    from megapath_amd import index_builder, simulate
    dev = torch.device("cuda", 0)
    n = 2_300_000_000
    text = simulate.genome(n, seed=11, device=dev, repeat_frac=0.0)      # no copied repeats: every error-free pair has exactly one home
    tc = text.cpu().numpy()
    cuts = np.linspace(0, n, 9).astype(np.int64)
    arrs = index_builder.build_index_arrays([("c%d" % i, tc[cuts[i]:cuts[i + 1]]) for i in range(8)], sa_interval=16, device=dev)
    del tc
    torch.cuda.empty_cache()
    gix = H.Index.from_arrays(0, **arrs)
    del arrs
    assert gix.info.prefilter_k == 17
    checked, bad = gix.check_lkt16(1 << 20)
    assert checked > n - 100 and bad == 0
    P, L = 1 << 17, 150
    codes, lens, truth = simulate.paired_reads(text, P, L, seed=5, sub=0.0, indel_frac=0.0, random_frac=0.1, return_truth=True)
    W = (L + 1 + 15) // 16
    packed = simulate.pack_codes(codes, W).cpu().numpy().view(np.uint32)
    lens_np = lens.cpu().numpy().view(np.uint32)
    params = H.AlignParams(L, 750, 130, 130, H.MmpParams.nt2())
    ctx = H.Context(gix)
    res = {}
    for on in (True, False):
        ctx.set_prefilter(on)
        ctx.set_reads(packed, lens_np)
        hits, nh, st = ctx.deep_dp_round(params)
        arr = np.frombuffer((C.c_char * (nh * 48)).from_address(C.addressof(hits)), dtype=taxcount_dtype(), count=nh).copy()
        res[on] = np.sort(arr, order=["pair_id", "algnmt_1", "algnmt_2", "score_1", "score_2"])
    assert len(res[True]) == len(res[False]) and (res[True] == res[False]).all()
    arr = res[True]
    pos = truth["pos"].cpu().numpy(); flen = truth["flen"].cpu().numpy(); flip = truth["flip"].cpu().numpy()
    rnd = (truth["random1"] | truth["random2"]).cpu().numpy()
    # the forward mate sits at the fragment's start, the reverse-strand mate at its end - L; a flipped pair has read 1 as the reverse one
    exp_left, exp_right = pos, pos + flen - L
    exp1, exp2 = np.where(flip, exp_right, exp_left), np.where(flip, exp_left, exp_right)
    ok = np.zeros(P, dtype=bool)
    is_exp = (arr["algnmt_1"] == exp1[arr["pair_id"]]) & (arr["algnmt_2"] == exp2[arr["pair_id"]]) & (arr["score_1"] == L) & (arr["score_2"] == L)
    ok[arr["pair_id"][is_exp]] = True
    assert ok[~rnd].all(), int((~ok[~rnd]).sum())
    assert not np.isin(np.nonzero(rnd)[0], arr["pair_id"]).any()
    assert (arr["algnmt_1"] > (1 << 31)).sum() > 1000 and int(arr["algnmt_1"].max()) > 2_290_000_000
    # the mate whose read is read 1 sits left exactly when the pair was not flipped
    assert ((arr["left_is_mate"][is_exp] == 1) == flip[arr["pair_id"][is_exp]]).all()
    ctx.close(); gix.close()


def taxcount_dtype():
    from megapath_amd import taxcount
    return taxcount.PAIR_HIT_DTYPE
