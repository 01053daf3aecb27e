"""-m gpu: the HIP path (through the C ABI of libmegapath_hip.so) against the CPU oracle and the committed golden
vectors of the real reference.  Bit-exact: integer / index work only."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import codes_of, oracle_seeds, read_fastq_pairs, to_oracle_mmp
from megapath_amd import hipapi as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gidx(g1_prefix):
    ix = H.Index(g1_prefix, 0)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def oidx(g1_prefix):
    lib = O.oracle()
    ix = lib.mpo_index_open(g1_prefix.encode())
    yield ix
    lib.mpo_index_close(ix)


def test_fm_primitives_match_reference_vectors(gidx, golden_dir):
    v = np.load(os.path.join(golden_dir, "fm_vectors.npz"))
    assert gidx.info.text_length == int(v["text_length"])
    assert np.array_equal(gidx.occ(v["occ_idx"], v["occ_c"]), v["occ"])
    assert np.array_equal(gidx.sa(v["sa_idx"]), v["sa"])
    l, r = gidx.lkt(v["lkt_key"])
    assert np.array_equal(l, v["lkt_range"][:, 0]) and np.array_equal(r, v["lkt_range"][:, 1])


def test_fm_primitives_exhaustive_small(gidx, oidx):
    lib = O.oracle()
    n = gidx.info.text_length
    idx = np.arange(0, n + 2, 7, dtype=np.uint64)
    for c in range(4):
        want = np.array([lib.mpo_occ(oidx, int(i), c) for i in idx], dtype=np.uint64)
        assert np.array_equal(gidx.occ(idx, np.full(len(idx), c, dtype=np.uint32)), want)
    sidx = np.arange(0, n + 1, 5, dtype=np.uint64)
    want = np.array([lib.mpo_sa_value(oidx, int(i)) for i in sidx], dtype=np.uint64)
    assert np.array_equal(gidx.sa(sidx), want)


@pytest.mark.parametrize("n_bases", [256 * 9 + 200, 192 * 25, 4096 + 17])
def test_fm_primitives_every_index_at_text_tail_shapes(tmp_path, n_bases, monkeypatch):
    monkeypatch.setenv("MP_SA_DENSE", "1" if n_bases % 2 else "0")        # the denser SA table derived at load time, on for one shape
    """Text lengths that end inside the upper half of an explicit-occurrence interval, on a rank-block boundary and just
    past a power of two: every rank (all four symbols) and every SA value, device vs oracle, on builder-written files."""
    from megapath_amd import index_builder, synth
    genome = synth.random_genome(n_bases, 78, n_seqs=3)
    prefix = str(tmp_path / "t.index")
    index_builder.write_index_files(prefix, index_builder.build_index_arrays(genome, sa_interval=16))
    lib = O.oracle()
    oix = lib.mpo_index_open(prefix.encode())
    gix = H.Index(prefix, 0)
    n = gix.info.text_length
    idx = np.arange(0, n + 2, dtype=np.uint64)
    for c in range(4):
        want = np.array([lib.mpo_occ(oix, int(i), c) for i in idx], dtype=np.uint64)
        assert np.array_equal(gix.occ(idx, np.full(len(idx), c, dtype=np.uint32)), want), c
    sidx = np.arange(0, n + 1, dtype=np.uint64)
    assert np.array_equal(gix.sa(sidx), np.array([lib.mpo_sa_value(oix, int(i)) for i in sidx], dtype=np.uint64))
    gix.close()
    lib.mpo_index_close(oix)


def test_kmer_prefilter_bitmap_equals_table_plus_four_steps(g1_prefix, monkeypatch):
    """The 17-mer pre-filter bit must say exactly what the reference procedure would find: 13-mer table range of the first
    13 scanned bases, then four backward-search steps (DV-DPfunctions.cpp:2245-2262).  Checked for every 17-base window
    of the text (incl. the windows that run into the table's 'A' padding at the text end) and for random 17-mers."""
    monkeypatch.setenv("MP_KMER_FILTER", "1")
    monkeypatch.setenv("MP_KMER15", "1")
    gix = H.Index(g1_prefix, 0)
    assert gix.info.prefilter_k == 17
    lib = O.oracle()
    oix = lib.mpo_index_open(g1_prefix.encode())
    n = gix.info.text_length
    text = np.array([lib.mpo_text_base(oix, i) for i in range(n)] + [0] * 16, dtype=np.uint64)
    starts = np.concatenate([np.arange(0, 3000), np.arange(n - 40, n)])            # windows start .. start+16, padded with 'A'
    keys = np.zeros(len(starts), dtype=np.uint64)
    for t in range(17):                                                            # leftmost text base in the top bits
        keys |= text[starts + t] << np.uint64(2 * (16 - t))
    rng = np.random.default_rng(5)
    keys = np.concatenate([keys, rng.integers(0, 1 << 34, size=20000, dtype=np.uint64)])
    got = gix.prefilter(keys)
    # the procedure, through the device's own table / rank queries (themselves pinned against the reference)
    cum = [int(gix.info.cum_freq[c]) for c in range(5)]
    l, r = gix.lkt((keys & np.uint64((1 << 26) - 1)).astype(np.uint32))
    l = l.astype(np.int64); r = r.astype(np.int64)
    for step in range(4):
        c = ((keys >> np.uint64(26 + 2 * step)) & np.uint64(3)).astype(np.uint32)
        alive = l <= r
        li = np.where(alive, l, 0).astype(np.uint64); ri = np.where(alive, r + 1, 0).astype(np.uint64)
        cl = np.array([cum[int(x)] for x in c], dtype=np.int64)
        nl = cl + gix.occ(li, c).astype(np.int64) + 1
        nr = cl + gix.occ(ri, c).astype(np.int64)
        l = np.where(alive, nl, 1); r = np.where(alive, nr, 0)
    want = (l <= r).astype(np.uint8)
    assert want[:3040].sum() >= 3030                      # text windows are present (those cut by the text end may not be)
    assert np.array_equal(got, want)
    # the per-15-mer arrangement of the same bits (one load per three probe positions): all three alignments agree
    got3 = gix.prefilter3(keys)
    assert np.array_equal(got3, want * 7)
    gix.close()
    lib.mpo_index_close(oix)


@pytest.mark.parametrize("n_bases", [600000, 192 * 25 + 3, 4096 + 17])
def test_lkt16_table_equals_table_plus_three_steps(tmp_path, n_bases, monkeypatch):
    """The 16-mer look-up table the seeding kernel reads behind the pre-filter must hold, for every key the kernel uses it
    for, the SA range the index's own procedure reaches: 13-mer table, then three backward-search steps.  Checked on the
    device for the 16-mer at every text position (incl. those running into the 'A' padding at the text end, and a text
    ending in a poly-A run) and for random keys."""
    from megapath_amd import index_builder, synth
    monkeypatch.setenv("MP_KMER_FILTER", "1")
    monkeypatch.setenv("MP_LKT16", "1")
    monkeypatch.setenv("MP_SA_DENSE", "1")
    monkeypatch.setenv("MP_LKT16U", "1")                  # the 16-byte entries {l, r, pos, prev16} are verified by the same kernel
    genome = synth.random_genome(n_bases, 91, n_seqs=2)
    if n_bases < 10000:                                   # a poly-A tail: the buckets the kernel leaves to the 13-mer path
        name, arr = genome[-1]
        arr = arr.copy(); arr[-20:] = ord("A")
        genome[-1] = (name, arr)
    gix = H.Index.from_arrays(0, **index_builder.build_index_arrays(genome, sa_interval=16))
    checked, bad = gix.check_lkt16(500000)
    assert bad == 0
    assert checked > n_bases - 200 + 400000
    gix.close()


@pytest.mark.parametrize("tag,mmp_name", [("a", "nt2"), ("c", "nt2"), ("a", "human"), ("b", "human"), ("c", "human")])
def test_seeds_and_hits_identical_with_prefilter(g1_prefix, golden_dir, monkeypatch, tag, mmp_name):
    """Same kept seeds and the same deep-DP hits with the 17-mer pre-filter (and the 16-mer table behind it) on and off;
    fewer rank queries with it.  With soap4.ini's 22-base seeds a start is ruled out by any of the six 17-mers inside its
    first 22 bases (seed_kernel), so the scan jumps over whole runs of starts: still the same seeds."""
    mmp = getattr(H.MmpParams, mmp_name)()
    monkeypatch.setenv("MP_KMER_FILTER", "1")
    monkeypatch.setenv("MP_KMER15", "1")
    monkeypatch.setenv("MP_LKT16", "1")
    monkeypatch.setenv("MP_SA_DENSE", "1")
    monkeypatch.setenv("MP_LKT16U", "1")
    gix = H.Index(g1_prefix, 0)
    d = os.path.join(golden_dir, "g1")
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    seqs = read_fastq_pairs(os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag))
    packed, lens, W = H.pack_reads(seqs, L)
    out = {}
    for on in (True, False):
        ctx = H.Context(gix)
        ctx.set_prefilter(on)
        ctx.set_reads(packed, lens)
        st = ctx.seed_batch(mmp)
        arr, off = ctx.get_seeds()
        seeds = [(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[-1]))]
        hits, nh, _ = ctx.deep_dp_round(H.AlignParams(max(150, int(lens.max())), 750, 130, 130, mmp, 1, 0))
        rec = sorted((hits[i].pair_id, hits[i].left_is_mate, hits[i].algnmt_1, hits[i].algnmt_2, hits[i].score_1, hits[i].score_2,
                      hits[i].start_1, hits[i].start_2, hits[i].count_1, hits[i].count_2) for i in range(nh))   # slot order is not fixed
        out[on] = (seeds, list(off), rec, st.n_occ, st.n_lkt, st.n_prefilter)
        ctx.close()
    assert out[True][:3] == out[False][:3]
    assert out[False][5] == 0 and out[True][5] > 0
    assert out[True][3] < out[False][3] and out[True][4] < out[False][4]
    gix.close()


@pytest.mark.parametrize("tag,mmp", [("a", H.MmpParams.nt2()), ("b", H.MmpParams.human()), ("c", H.MmpParams.nt2())])
def test_seed_stage_matches_oracle(gidx, oidx, golden_dir, tag, mmp, monkeypatch):
    if tag == "b":
        monkeypatch.setenv("MP_SEED_SLAB_LIMIT", "2")     # nearly every read goes through the overflow kernel
    d = os.path.join(golden_dir, "g1")
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    seqs = read_fastq_pairs(os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag))
    packed, lens, W = H.pack_reads(seqs, L)
    ctx = H.Context(gidx)
    ctx.set_reads(packed, lens)
    st = ctx.seed_batch(mmp)
    arr, off = ctx.get_seeds()
    want, ctr = oracle_seeds(oidx, seqs, L, to_oracle_mmp(mmp))
    got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))]
           for r in range(len(seqs))]
    assert got == want
    # event counters feed the roofline: they must equal the oracle's count of the same events
    assert (st.n_occ, st.n_lkt, st.n_sa, st.n_seeds, st.n_hits) == (ctr.n_occ, ctr.n_lkt, ctr.n_sa, ctr.n_seeds, ctr.n_hits)
    ctx.close()


@pytest.mark.parametrize("tag,mmp,sa_dense", [("a", H.MmpParams.nt2(), "0"), ("b", H.MmpParams.human(), "0"), ("c", H.MmpParams.nt2(), "1")])
def test_seed_stage_with_the_16mer_filter_matches_oracle(oidx, golden_dir, tag, mmp, sa_dense, monkeypatch):
    """MP_KMER_FILTER=16: one bit per 16-mer (512 MiB) instead of per 17-mer (2 GiB) in front of every seed start -- the filter of the lean
    layout (many chunks resident).  A weaker filter, the same seeds: equal to the oracle's, with the overflow kernel (tag b) and with the
    whole suffix array resident (text comparison path)."""
    for k in ("MP_KMER15", "MP_LKT16", "MP_LKT16U"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("MP_KMER_FILTER", "16")
    monkeypatch.setenv("MP_SA_DENSE", sa_dense)
    if tag == "b":
        monkeypatch.setenv("MP_SEED_SLAB_LIMIT", "2")
    d = os.path.join(golden_dir, "g1")
    gix = H.Index(os.path.join(d, "ref.fa.index"), 0)
    assert gix.info.prefilter_k == 16
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    seqs = read_fastq_pairs(os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag))
    packed, lens, W = H.pack_reads(seqs, L)
    ctx = H.Context(gix)
    ctx.set_reads(packed, lens)
    st = ctx.seed_batch(mmp)
    arr, off = ctx.get_seeds()
    want, ctr = oracle_seeds(oidx, seqs, L, to_oracle_mmp(mmp))
    got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))] for r in range(len(seqs))]
    assert got == want
    assert st.n_prefilter > 0 and (st.n_seeds, st.n_hits) == (ctr.n_seeds, ctr.n_hits)
    ctx.close()
    gix.close()


@pytest.mark.parametrize("tag,mmp", [("a", H.MmpParams.nt2()), ("b", H.MmpParams.human()), ("c", H.MmpParams.nt2())])
def test_seed_stage_with_the_packed_14mer_table_matches_oracle(gidx, oidx, golden_dir, tag, mmp, monkeypatch):
    """MP_LKT14=1: the 13-mer table as 8-byte entries that carry a seed's first extension step (512 MiB in place of 256; the lean layout of many
    resident chunks).  The same seeds as the oracle's, one rank query pair less per seed start; the 13-mer ranges read back out of the entries
    equal the plain table's; with the overflow kernel (tag b), which takes the 13-mer path through the entries."""
    for k in ("MP_KMER_FILTER", "MP_KMER15", "MP_LKT16", "MP_LKT16U", "MP_SA_DENSE"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("MP_LKT14", "1")
    if tag == "b":
        monkeypatch.setenv("MP_SEED_SLAB_LIMIT", "2")
    d = os.path.join(golden_dir, "g1")
    gix = H.Index(os.path.join(d, "ref.fa.index"), 0)

    rng = np.random.default_rng(11)
    keys = np.concatenate([np.array([0, 1, (1 << 26) - 2, (1 << 26) - 1], dtype=np.uint32), rng.integers(0, 1 << 26, size=200000, dtype=np.uint32)])
    la, ra = gix.lkt(keys)
    lb, rb = gidx.lkt(keys)
    assert np.array_equal(la, lb) and np.array_equal(ra, rb)
    assert np.array_equal(gix.lkt_table(), gidx.lkt_table())
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    seqs = read_fastq_pairs(os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag))
    packed, lens, W = H.pack_reads(seqs, L)
    ctx = H.Context(gix)
    ctx.set_reads(packed, lens)
    st = ctx.seed_batch(mmp)
    arr, off = ctx.get_seeds()
    want, ctr = oracle_seeds(oidx, seqs, L, to_oracle_mmp(mmp))
    got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))] for r in range(len(seqs))]
    assert got == want
    assert (st.n_lkt, st.n_seeds, st.n_hits) == (ctr.n_lkt, ctr.n_seeds, ctr.n_hits)
    assert st.n_occ < ctr.n_occ and st.n_steps_skipped > 0
    ctx.close()
    gix.close()


def test_packed_14mer_table_on_a_repeat_rich_genome(tmp_path, monkeypatch):
    """Entries of the packed table whose counts do not fit a byte (poly-A, two-base and longer tandem repeats: 14-mers with hundreds of
    occurrences) say "no short cut" and the seed start takes the 13-mer path through the same entries: seeds equal to the oracle's, and both
    kinds of entry are met."""
    from megapath_amd import index_builder, synth
    import oracle_lib as O
    genome = synth.random_genome(400000, 3301, n_seqs=2, repeat_frac=0.2)
    rng = np.random.Generator(np.random.PCG64(3302))
    g2 = []
    for name, arr in genome:
        arr = arr.copy()
        for _ in range(40):                                   # long low-complexity stretches: homopolymers and short-unit tandem repeats
            u = int(rng.integers(1, 7)); ln = int(rng.integers(400, 3000)); p = int(rng.integers(0, len(arr) - ln - u))
            unit = arr[p:p + u].copy()
            arr[p:p + ln] = np.resize(unit, ln)
        g2.append((name, arr))
    arrs = index_builder.build_index_arrays(g2, sa_interval=16)
    prefix = str(tmp_path / "r.index")
    index_builder.write_index_files(prefix, arrs)
    for k in ("MP_KMER_FILTER", "MP_KMER15", "MP_LKT16", "MP_LKT16U", "MP_SA_DENSE"):
        monkeypatch.setenv(k, "0")
    plain = H.Index(prefix, 0)
    monkeypatch.setenv("MP_LKT14", "1")
    gix = H.Index(prefix, 0)
    keys = np.arange(0, 1 << 26, 97, dtype=np.uint32)
    la, ra = gix.lkt(keys)
    lb, rb = plain.lkt(keys)
    assert np.array_equal(la, lb) and np.array_equal(ra, rb)
    names, r1, r2 = synth.simulate_pairs(g2, 3000, 150, 3303, sub=0.01, ins=0.001, dele=0.001, random_frac=0.05)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, 151)
    mmp = H.MmpParams.nt2()
    lib = O.oracle()
    oix = lib.mpo_index_open(prefix.encode())
    want, ctr = oracle_seeds(oix, seqs, 151, to_oracle_mmp(mmp))
    lib.mpo_index_close(oix)
    stats = []
    for ix in (gix, plain):
        ctx = H.Context(ix)
        ctx.set_reads(packed, lens)
        st = ctx.seed_batch(mmp)
        arr, off = ctx.get_seeds()
        got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))] for r in range(len(seqs))]
        assert got == want
        assert (st.n_seeds, st.n_hits) == (ctr.n_seeds, ctr.n_hits)
        stats.append(st)
        ctx.close()
    # look-ups: one per seed start with the plain table; with the entries one per start plus one more for each start that met a "no short cut" entry
    assert stats[0].n_lkt > stats[1].n_lkt == ctr.n_lkt and stats[0].n_steps_skipped > 0 and stats[0].n_occ < stats[1].n_occ
    gix.close(); plain.close()


@pytest.mark.parametrize("dense", [False, True])
def test_merge_kernel_paths_that_do_not_stage_in_lds(golden_dir, oidx, monkeypatch, dense):
    """merge_kernel handles a block of 64 reads whose hits do not fit in its 32 KB of LDS one lane per read in global memory.  Forced for every
    block (MP_MERGE_STAGE_CAP=0), with the keys made from the seed records (whole suffix array resident) and from the located hits:
    the kept seeds must equal the oracle's."""
    monkeypatch.setenv("MP_MERGE_STAGE_CAP", "0")
    monkeypatch.setenv("MP_SA_DENSE", "1" if dense else "0")
    for k in ("MP_KMER_FILTER", "MP_KMER15", "MP_LKT16", "MP_LKT16U"):
        monkeypatch.setenv(k, "1" if dense else "0")
    d = os.path.join(golden_dir, "g1")
    gix = H.Index(os.path.join(d, "ref.fa.index"), 0)
    seqs = read_fastq_pairs(os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"))
    packed, lens, W = H.pack_reads(seqs, 151)
    ctx = H.Context(gix)
    ctx.set_reads(packed, lens)
    mmp = H.MmpParams.nt2()
    ctx.seed_batch(mmp)
    arr, off = ctx.get_seeds()
    want, _ = oracle_seeds(oidx, seqs, 151, to_oracle_mmp(mmp))
    got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))] for r in range(len(seqs))]
    assert got == want
    ctx.close()
    gix.close()


@pytest.mark.parametrize("rows", ["1", "0"])
def test_merge_rows_and_whole_wavefronts_equal_oracle(tmp_path, monkeypatch, rows):
    """merge_kernel gives a read with at most 16 located hits one 16-lane row of a wavefront (four reads at a time, DPP row rotations) and
    a read with more the whole wavefront; MP_MERGE_ROWS=0 sends every read the wave-wide way.  A repeat-rich genome (reads with 1, a few,
    17 .. 64 and more than 64 hits side by side in one quad of reads, groups of several hits within indelFuzz through tandem copies):
    the kept seeds equal the oracle's either way."""
    from megapath_amd import index_builder, synth
    import oracle_lib as O
    monkeypatch.setenv("MP_MERGE_ROWS", rows)
    genome = synth.random_genome(300000, 2201, n_seqs=2, repeat_frac=0.35)
    rng = np.random.Generator(np.random.PCG64(2202))
    g2 = []
    for name, arr in genome:
        arr = arr.copy()
        for _ in range(300):
            u = int(rng.integers(3, 30)); k = int(rng.integers(2, 6)); p = int(rng.integers(0, len(arr) - u * (k + 1)))
            for c in range(1, k + 1):
                arr[p + c * u:p + (c + 1) * u] = arr[p:p + u]
        g2.append((name, arr))
    arrs = index_builder.build_index_arrays(g2, sa_interval=16)
    prefix = str(tmp_path / "m.index")
    index_builder.write_index_files(prefix, arrs)
    gix = H.Index(prefix, 0)
    names, r1, r2 = synth.simulate_pairs(g2, 2500, 150, 2203, sub=0.01, ins=0.001, dele=0.001, random_frac=0.02)
    seqs = [x for pr in zip(r1, r2) for x in pr]
    packed, lens, W = H.pack_reads(seqs, 151)
    ctx = H.Context(gix)
    ctx.set_reads(packed, lens)
    mmp = H.MmpParams.nt2()
    st = ctx.seed_batch(mmp)
    arr, off = ctx.get_seeds()
    lib = O.oracle()
    oix = lib.mpo_index_open(prefix.encode())
    want, ctr = oracle_seeds(oix, seqs, 151, to_oracle_mmp(mmp))
    lib.mpo_index_close(oix)
    got = [[(arr[i].pos, arr[i].strand_read_id, arr[i].paired_seed_length) for i in range(int(off[r]), int(off[r + 1]))] for r in range(len(seqs))]
    assert got == want
    per_read = np.diff(off)
    assert (per_read == 1).sum() > 200 and (per_read > 4).sum() > 200        # single hits and crowded reads are both there
    assert st.n_seeds == ctr.n_seeds and st.n_hits == ctr.n_hits and st.n_hits > 17 * len(seqs) // 4
    ctx.close()
    gix.close()


def test_dp_matches_reference_vectors(golden_dir, tmp_path):
    """callDP golden vectors: build a throw-away index whose text is the concatenation of the windows."""
    v = np.load(os.path.join(golden_dir, "dp_vectors.npz"))
    ro = np.concatenate([[0], np.cumsum(v["ref_len"])])
    qo = np.concatenate([[0], np.cumsum(v["read_len"])])
    text = v["ref_cat"].astype(np.uint8)
    from megapath_amd.index_builder import build_index_arrays
    arrs = build_index_arrays([("cat", text)], sa_interval=16)
    ix = H.Index.from_arrays(0, **arrs)
    ctx = H.Context(ix)
    clips = sorted(set(map(tuple, v["clip"].tolist())))
    n_checked = 0
    for clip in clips:
        sel = [t for t in range(len(v["score"])) if tuple(v["clip"][t]) == clip]
        seqs = []
        for t in sel:
            s = bytes(b"ACGT"[c] for c in v["read_cat"][qo[t]:qo[t + 1]])
            seqs += [s, s]
        L = max(len(s) for s in seqs) + 1
        packed, lens, W = H.pack_reads(seqs, L)
        ctx.set_reads(packed, lens)
        tasks = [H.DpTask(2 * k, 1, int(ro[t]), int(v["ref_len"][t]), 0) for k, t in enumerate(sel)]
        res, ops, stride, _ = ctx.sgdp_batch(tasks, clip[0], clip[1], want_ops=True)
        for k, t in enumerate(sel):
            cig = ""
            if res[k].score > 0:
                buf = C.create_string_buffer(4 * stride + 16)
                o = np.ascontiguousarray(ops[k * stride:k * stride + res[k].n_ops])
                O.oracle().mpo_cigar_from_ops(O.as_u8p(o), res[k].n_ops, buf, len(buf))
                cig = buf.value.decode()
            got = (res[k].score, res[k].hit_loc, res[k].max_score_count, cig)
            want = (int(v["score"][t]), int(v["hit_loc"][t]), int(v["count"][t]), str(v["cigar"][t]))
            assert got == want, (t, clip)
            n_checked += 1
    assert n_checked == len(v["score"])
    ctx.close()
    ix.close()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_deep_dp_round_matches_reference_records(gidx, oidx, golden_dir, tmp_path, tag):
    """The deep-DP round on the GPU must reproduce, for every pair the oracle aligns in that round, the same best pair
    (positions + scores); checked through the oracle's record dump and through the reference's expected stdout."""
    lib = O.oracle()
    d = os.path.join(golden_dir, "g1")
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    fq1, fq2 = os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag)
    p = O.default_params(L=L)
    st = O.AlignStats()
    rec = tmp_path / "rec.tsv"
    assert lib.mpo_align_fastq(oidx, C.byref(p), fq1.encode(), fq2.encode(), str(tmp_path / "o.fq").encode(), str(rec).encode(), C.byref(st)) == 0
    want = {}
    for line in open(rec):
        name, mate, stage, seq, pos, strand, score, cigar = line.rstrip("\n").split("\t")
        if stage == "deep":
            want.setdefault(name, {})[int(mate)] = (seq, int(pos), int(score))
    seqs = read_fastq_pairs(fq1, fq2)
    names = [l[1:].split()[0][:-2] for l in open(fq1) if l.startswith("@r")]
    packed, lens, W = H.pack_reads(seqs, L)
    ctx = H.Context(gidx)
    ctx.set_reads(packed, lens)
    lens1 = lens[0::2]
    lens2 = lens[1::2]
    ap = H.AlignParams(max(1, int(lens1.max()), int(lens2.max())), 750, 130, 130, H.MmpParams.nt2())
    hits, nh, rst = ctx.deep_dp_round(ap)
    assert rst.n_candidates == st.n_candidates
    by_pair = {}
    for i in range(nh):
        h = hits[i]
        by_pair.setdefault(h.pair_id, set()).add((h.algnmt_1, h.algnmt_2, h.score_1, h.score_2))
    assert len(by_pair) == st.n_deep_dp_pairs == len(want)
    for pid, hs in by_pair.items():
        best = None
        for h in sorted(hs):                      # ResultCompare order, first maximum wins
            if best is None or h[2] + h[3] > best[2] + best[3]:
                best = h
        w = want[names[pid]]
        p1, c1 = gidx.chr_and_pos(best[0])
        p2, c2 = gidx.chr_and_pos(best[1])
        assert (gidx.seq_name(c1), p1, best[2]) == w[1]
        assert (gidx.seq_name(c2), p2, best[3]) == w[2]
    ctx.close()


def test_device_longer_seed_sort_is_the_reference_order(golden_dir):
    """single_cand_kernel's candidate order (std::sort with LongerSeedLength, DV-DPfunctions.cpp:334): recorded libstdc++
    permutations and the oracle on random tie-rich arrays."""
    lib = O.oracle()
    v = np.load(os.path.join(golden_dir, "sort_vectors.npz"))
    lists = [v["seed_len"][int(v["off"][a]):int(v["off"][a + 1])] for a in range(len(v["off"]) - 1)]
    got = H.sort_longer_seed(lists)
    for a, g in enumerate(got):
        assert np.array_equal(g.astype(np.int64), v["perm"][int(v["off"][a]):int(v["off"][a + 1])].astype(np.int64)), a
    rng = np.random.default_rng(11)
    lists = [(17 + rng.integers(0, int(rng.integers(1, 9)), size=int(rng.integers(0, 1500)))).astype(np.uint32) for _ in range(400)]
    got = H.sort_longer_seed(lists)
    for a, (lens, g) in enumerate(zip(lists, got)):
        perm = np.zeros(max(len(lens), 1), dtype=np.int32)
        lib.mpo_sort_longer_seed_perm(np.ascontiguousarray(lens).ctypes.data, len(lens), perm.ctypes.data)
        assert np.array_equal(g.astype(np.int64), perm[:len(lens)].astype(np.int64)), a


# ------------------------------------------------------------------------------------------------ native index builder (8f row 1)
def _fasta_seqs(path):
    seqs, name, buf = [], None, []
    for line in open(path, "rb"):
        line = line.rstrip()
        if line.startswith(b">"):
            if name is not None:
                seqs.append((name, np.frombuffer(b"".join(buf), dtype=np.uint8)))
            name, buf = line[1:].split()[0].decode(), []
        else:
            buf.append(line)
    seqs.append((name, np.frombuffer(b"".join(buf), dtype=np.uint8)))
    return seqs


def test_native_builder_writes_the_reference_builders_files(golden_dir, tmp_path):
    """mp_build_index_arrays (prefix doubling on the device) + the host file writers against the files the REAL 2bwt-builder wrote:
    tests/golden/g1 (three sequences, 8 % copied repeats) and tests/golden/nrun (N runs removed, IUPAC codes) -- .bwt .fmv .sa .pac
    .tra .amb byte-identical."""
    from megapath_amd.index_builder import build_index_arrays, parse_fasta_reference_style, write_index_files
    d = os.path.join(golden_dir, "g1")
    arrs = build_index_arrays(_fasta_seqs(os.path.join(d, "ref.fa")), sa_interval=16, native=True)
    prefix = str(tmp_path / "g1.index")
    write_index_files(prefix, arrs, write_lkt=False)
    for ext in (".bwt", ".sa", ".pac", ".fmv"):
        assert open(prefix + ext, "rb").read() == open(os.path.join(d, "ref.fa.index" + ext), "rb").read(), ext
    d = os.path.join(golden_dir, "nrun")
    arrs = build_index_arrays(parse_fasta_reference_style(open(os.path.join(d, "n.fa"), "rb").read()), sa_interval=16, native=True)
    prefix = str(tmp_path / "n.index")
    write_index_files(prefix, arrs, write_lkt=False)
    for ext in (".pac", ".bwt", ".fmv", ".sa", ".tra", ".amb"):
        assert open(prefix + ext, "rb").read() == open(os.path.join(d, "n.fa.index" + ext), "rb").read(), ext


@pytest.mark.parametrize("n,repeat,interval", [(1, 0.0, 1), (17, 0.0, 4), (5000, 0.0, 16), (300000, 0.5, 16), (2000003, 0.3, 32)])
def test_native_builder_equals_the_torch_restatement(n, repeat, interval):
    """Same arrays from the HIP builder and from the torch prefix doubling (which the not-gpu tests pin on the real builder's files),
    on texts of awkward lengths, with long copied repeats (many doubling rounds) and a homopolymer tail."""
    from megapath_amd import index_builder, synth
    rng = np.random.default_rng(n)
    genome = synth.random_genome(n, 1000 + n, n_seqs=1, repeat_frac=repeat) if n >= 5000 else [("s", rng.integers(0, 4, n, dtype=np.uint8))]
    if n >= 5000:
        genome[0][1][-700:] = ord("G")                            # a run of one letter at the very end: the '$' ordering decides
    a = index_builder.build_index_arrays(genome, sa_interval=interval, native=True)
    rounds = index_builder.last_build_info["rounds"]
    b = index_builder.build_index_arrays(genome, sa_interval=interval, native=False, device="cpu")
    assert a["n"] == b["n"] and a["inv_sa0"] == b["inv_sa0"]
    for k in ("cum", "bwt_words", "sa_samples", "pac"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    if repeat > 0:
        assert rounds >= 5                                        # kilobase repeats need h up to a few thousand


@pytest.mark.parametrize("kind", ["one_letter", "period_2", "period_3_then_random"])
def test_native_builder_on_degenerate_texts(kind):
    """The worst cases of prefix doubling -- a single letter, a short period (suffixes differ only at the very end: log2(n) rounds) --
    against the torch restatement."""
    from megapath_amd import index_builder
    n = 70001
    rng = np.random.default_rng(3)
    if kind == "one_letter":
        codes = np.full(n, 3, dtype=np.uint8)
    elif kind == "period_2":
        codes = np.tile(np.array([1, 2], dtype=np.uint8), n // 2 + 1)[:n]
    else:
        codes = np.concatenate([np.tile(np.array([0, 0, 3], dtype=np.uint8), 20000), rng.integers(0, 4, n - 60000, dtype=np.uint8)])
    a = index_builder.build_index_arrays([("s", codes)], sa_interval=16, native=True)
    rounds = index_builder.last_build_info["rounds"]
    b = index_builder.build_index_arrays([("s", codes)], sa_interval=16, native=False, device="cpu")
    assert a["inv_sa0"] == b["inv_sa0"]
    for k in ("cum", "bwt_words", "sa_samples", "pac"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert rounds >= 10
