"""CPU tests: the oracle (oracle/*.c) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py) -- SURVEY.md section 8(c).  No GPU, no /root/reference needed."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import fastq_records


@pytest.fixture(scope="module")
def g1(g1_prefix):
    lib = O.oracle()
    ix = lib.mpo_index_open(g1_prefix.encode())
    yield ix
    lib.mpo_index_close(ix)


def test_fm_primitives_match_reference_vectors(g1, golden_dir):
    lib = O.oracle()
    v = np.load(os.path.join(golden_dir, "fm_vectors.npz"))
    assert lib.mpo_text_length(g1) == int(v["text_length"])
    assert lib.mpo_inverse_sa0(g1) == int(v["inverse_sa0"])
    occ = [lib.mpo_occ(g1, int(i), int(c)) for i, c in zip(v["occ_idx"], v["occ_c"])]
    assert np.array_equal(np.array(occ, dtype=np.uint64), v["occ"])
    sa = [lib.mpo_sa_value(g1, int(i)) for i in v["sa_idx"]]
    assert np.array_equal(np.array(sa, dtype=np.uint64), v["sa"])
    a, b = C.c_uint64(), C.c_uint64()
    for k, (l, r) in zip(v["lkt_key"], v["lkt_range"]):
        lib.mpo_lkt_range(g1, int(k), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (int(l), int(r))


def test_dp_matches_reference_vectors(golden_dir):
    v = np.load(os.path.join(golden_dir, "dp_vectors.npz"))
    ro = np.concatenate([[0], np.cumsum(v["ref_len"])])
    qo = np.concatenate([[0], np.cumsum(v["read_len"])])
    n_pass = 0
    for t in range(len(v["score"])):
        ref = v["ref_cat"][ro[t]:ro[t + 1]]
        read = v["read_cat"][qo[t]:qo[t + 1]]
        got = O.sgdp(ref, read, int(v["clip"][t][0]), int(v["clip"][t][1]))
        want = (int(v["score"][t]), int(v["hit_loc"][t]), int(v["count"][t]), str(v["cigar"][t]))
        assert got == want, t
        n_pass += want[0] > 0
    assert n_pass > 100


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_whole_path_matches_reference_stdout(g1, golden_dir, tmp_path, tag):
    lib = O.oracle()
    d = os.path.join(golden_dir, "g1")
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    p = O.default_params(L=L, insert_high=750, mode=1, top=95)
    st = O.AlignStats()
    out = tmp_path / "out.fq"
    rc = lib.mpo_align_fastq(g1, C.byref(p), os.path.join(d, "reads_%s_1.fq" % tag).encode(),
                             os.path.join(d, "reads_%s_2.fq" % tag).encode(), str(out).encode(), None, C.byref(st))
    assert rc == 0
    got = fastq_records(str(out))
    want = fastq_records(os.path.join(d, "expected_%s.fq" % tag))
    assert len(got) == len(want) == 2 * st.n_pairs
    assert got == want
    # every stage of the path must have been exercised by the fixture
    assert st.n_deep_dp_pairs > 0 and st.n_default_dp_pairs > 0 and st.n_unpaired_pairs > 0


@pytest.mark.parametrize("tag,reads,mode,nc", [("d", "d", 2, False), ("f", "d", 2, True), ("e", "e", 1, False)])
def test_paired_only_ignore_and_chained_chunk_goldens(g1, golden_dir, tmp_path, tag, reads, mode, nc):
    """"d": -P with IGNORE comments; "f": the same with -nc (comments dropped first); "e": the second chunk of the NT loop, header scores of the first chunk merged.
    Expected files are the real reference's stdout (tests/golden/make_golden.py)."""
    lib = O.oracle()
    d = os.path.join(golden_dir, "g1")
    p = O.default_params(L=151, insert_high=750, mode=mode, top=95, nc=nc)
    st = O.AlignStats()
    out = tmp_path / "out.fq"
    rc = lib.mpo_align_fastq(g1, C.byref(p), os.path.join(d, "reads_%s_1.fq" % reads).encode(), os.path.join(d, "reads_%s_2.fq" % reads).encode(),
                             str(out).encode(), None, C.byref(st))
    assert rc == 0
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_%s.fq" % tag))


# (set, -L, MegaPath mode, -nc, -p): the flags the goldens were made with (tests/golden/make_golden_bam.py)
BAM_SETS = {"a": (151, 1, True, False), "c": (301, 1, True, False), "d": (151, 2, False, False), "g": (151, 1, False, True),
            "h": (151, 1, False, False)}


@pytest.mark.parametrize("tag", sorted(BAM_SETS))
def test_bam_records_match_reference_bam(g1_prefix, golden_dir, tmp_path, tag):
    """Row a15: every BAM record the real `soap4 -b` wrote (.dpout.1 / .gout.1 / .unpair, decoded), field for field:
    flag, position, MAPQ, CIGAR, mate fields, tlen, sequence, qualities and the RG X0 X1 XM XO XG [NM MD] XA PH tags."""
    from conftest import bam_golden
    L, mode, nc, mdnm = BAM_SETS[tag]
    g = os.path.join(golden_dir, "g1")
    P = O.default_params(L=L, mode=mode, nc=nc)
    got = O.align_fastq_bam(g1_prefix, P, os.path.join(g, "reads_%s_1.fq" % tag), os.path.join(g, "reads_%s_2.fq" % tag),
                            str(tmp_path / "o.fq"), str(tmp_path / "o"), print_mdnm=mdnm, read_group="reads_%s_1.fq" % tag)
    want = bam_golden(os.path.join(g, "expected_%s.bam.txt" % tag))
    for k in ("dpout", "gout", "unpair"):
        assert got[k] == want[k], k
    assert want["header"].startswith(got["header"])          # the reference appends its own @PG line
    assert open(tmp_path / "o.fq", "rb").read() == open(os.path.join(g, "expected_%s.fq" % tag), "rb").read()


# ------------------------------------------------------------------------------------------------ g2: corners g1 does not reach
def test_longer_seed_sort_is_libstdcxx_introsort(golden_dir):
    """singleMerge's std::sort(.., LongerSeedLength()) (DV-DPfunctions.cpp:334): the oracle's restatement of libstdc++'s
    introsort against permutations recorded from the real std::sort (tests/golden/make_golden_sort.py)."""
    lib = O.oracle()
    v = np.load(os.path.join(golden_dir, "sort_vectors.npz"))
    n_big = 0
    for a in range(len(v["off"]) - 1):
        lo, hi = int(v["off"][a]), int(v["off"][a + 1])
        lens = np.ascontiguousarray(v["seed_len"][lo:hi], dtype=np.uint32)
        perm = np.zeros(max(hi - lo, 1), dtype=np.int32)
        lib.mpo_sort_longer_seed_perm(lens.ctypes.data, hi - lo, perm.ctypes.data)
        assert np.array_equal(perm[:hi - lo], v["perm"][lo:hi]), a
        n_big += hi - lo > 16 and not np.array_equal(perm[:hi - lo], np.argsort(-lens.astype(np.int64), kind="stable"))
    assert n_big > 20          # the fixture does contain arrays a stable sort orders differently


@pytest.mark.parametrize("tag,L", [("t", 301), ("s", 151)])
def test_g2_tied_candidates_and_unseedable_reads(golden_dir, tmp_path, tag, L):
    """g2/t: reads with > 200 single-end candidates of equal seed length -- which 200 survive the cut
    (DV-DPForSingleReads.cpp:200) depends on std::sort's tie order; g2/s: mates of 1..16 bases."""
    lib = O.oracle()
    d = os.path.join(golden_dir, "g2")
    ix = lib.mpo_index_open(os.path.join(d, "ref.fa.index").encode())
    p = O.default_params(L=L, insert_high=750, mode=1, top=95)
    st = O.AlignStats()
    out = tmp_path / "out.fq"
    rc = lib.mpo_align_fastq(ix, C.byref(p), os.path.join(d, "reads_%s_1.fq" % tag).encode(), os.path.join(d, "reads_%s_2.fq" % tag).encode(),
                             str(out).encode(), None, C.byref(st))
    lib.mpo_index_close(ix)
    assert rc == 0
    got, want = fastq_records(str(out)), fastq_records(os.path.join(d, "expected_%s.fq" % tag))
    assert got == want
    if tag == "t":
        capped = [r for r in want if r[0].startswith("f") and r[2].count(",") == 200]
        assert len(capped) == 10       # every tied read was cut at 200 candidates


@pytest.mark.parametrize("tag,L", [("a", 151), ("c", 301)])
def test_three_chained_chunks_match_reference_chain(golden_dir, tmp_path, tag, L):
    """g3: the NT loop (runMegaPath.sh:191-227) over three index chunks, three chained reference runs; the oracle is chained
    the same way -- each run reads the previous run's de-interleaved stdout and merges its SCORE: comments."""
    lib = O.oracle()
    g1, g3 = os.path.join(golden_dir, "g1"), os.path.join(golden_dir, "g3")
    in1, in2 = os.path.join(g1, "reads_%s_1.fq" % tag), os.path.join(g1, "reads_%s_2.fq" % tag)
    out = None
    for k in range(3):
        ix = lib.mpo_index_open(os.path.join(g3, "c%d.fa.index" % k).encode())
        p = O.default_params(L=L, insert_high=750, mode=1, top=95, nc=(k == 0))
        st = O.AlignStats()
        out = str(tmp_path / ("chain_%d.fq" % k))
        assert lib.mpo_align_fastq(ix, C.byref(p), in1.encode(), in2.encode(), out.encode(), None, C.byref(st)) == 0
        lib.mpo_index_close(ix)
        lines = open(out, "rb").read().split(b"\n")
        in1, in2 = out + ".1", out + ".2"
        with open(in1, "wb") as a, open(in2, "wb") as b:
            for i in range(0, len(lines) - 7, 8):
                a.write(b"\n".join(lines[i:i + 4]) + b"\n")
                b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")
    got, want = fastq_records(out), fastq_records(os.path.join(g3, "expected_chain_%s.fq" % tag))
    assert got == want
    assert sum(1 for r in want if r[2].count(",") >= 2) > 0 or tag == "c"      # some reads do carry hits from more than one chunk


def test_nine_chained_chunks_match_reference_chain(golden_dir, tmp_path):
    """g6: nine index chunks with homologous sequences at 0-7 % divergence in several of them; nine chained runs of the real soap4
    (tests/golden/make_golden_g6.py).  The oracle chained the same way must end with the same records: the -top 95 rule decides chunk
    after chunk which per-reference scores stay in the header."""
    lib = O.oracle()
    g6 = os.path.join(golden_dir, "g6")
    in1, in2 = os.path.join(g6, "reads_1.fq"), os.path.join(g6, "reads_2.fq")
    out = None
    for k in range(9):
        ix = lib.mpo_index_open(os.path.join(g6, "c%d.fa.index" % k).encode())
        p = O.default_params(L=151, insert_high=750, mode=1, top=95, nc=(k == 0))
        st = O.AlignStats()
        out = str(tmp_path / ("chain_%d.fq" % k))
        assert lib.mpo_align_fastq(ix, C.byref(p), in1.encode(), in2.encode(), out.encode(), None, C.byref(st)) == 0
        lib.mpo_index_close(ix)
        lines = open(out, "rb").read().split(b"\n")
        in1, in2 = out + ".1", out + ".2"
        with open(in1, "wb") as a, open(in2, "wb") as b:
            for i in range(0, len(lines) - 7, 8):
                a.write(b"\n".join(lines[i:i + 4]) + b"\n")
                b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")
    got, want = fastq_records(out), fastq_records(os.path.join(g6, "expected_chain.fq"))
    assert got == want
    assert sum(1 for r in want if r[2].count(",c") >= 2) > 50


# (tag, MegaPath mode, -top, seed_min_length, reseed_len): tests/golden/g5, the stages that run with soap4.ini (runMegaPath.sh:136,162)
G5_SETS = {"h": (1, 95, 22, 23), "r": (2, 100, 22, 23), "n": (1, 95, 17, 18)}


@pytest.mark.parametrize("tag", sorted(G5_SETS))
def test_soap4_ini_stage_goldens(g1, golden_dir, tmp_path, tag):
    """Host filter (`-C soap4.ini -L 150 -F -nc`) and 16S filter (`-C soap4.ini -L 150 -F -P -nc -top 100`) as the pipeline runs them:
    mmpSeedMinLength 22 / mmpReseedLen 23 instead of the NT stage's 17 / 18.  Expected = stdout of the real soap4
    (tests/golden/make_golden_ini.py); "n" is the same read set with soap4-nt2.ini and differs from "h" in ten records."""
    lib = O.oracle()
    d = os.path.join(golden_dir, "g5")
    mode, top, smin, rlen = G5_SETS[tag]
    p = O.default_params(L=150, insert_high=750, mode=mode, top=top, nc=True)
    p.mmp.seed_min_length, p.mmp.reseed_len = smin, rlen
    st = O.AlignStats()
    out = tmp_path / "out.fq"
    assert lib.mpo_align_fastq(g1, C.byref(p), os.path.join(d, "reads_1.fq").encode(), os.path.join(d, "reads_2.fq").encode(), str(out).encode(), None, C.byref(st)) == 0
    got, want = fastq_records(str(out)), fastq_records(os.path.join(d, "expected_%s.fq" % tag))
    assert len(got) == len(want) == 2 * st.n_pairs and got == want
    assert fastq_records(os.path.join(d, "expected_h.fq")) != fastq_records(os.path.join(d, "expected_n.fq"))


@pytest.mark.parametrize("db", ["g3", "g6"])
def test_chained_chunks_with_bam_match_reference_chain(golden_dir, tmp_path, db):
    """The NT loop as runMegaPath.sh:191-227 really runs it: EVERY run gets -b and the same -o prefix, so the BAM files that remain are the last
    chunk's (tests/golden/make_golden_g7.py, from the real soap4).  The oracle chained the same way: the last run's stdout and its BAM records."""
    from conftest import bam_golden
    lib = O.oracle()
    g1, d = os.path.join(golden_dir, "g1"), os.path.join(golden_dir, db)
    n = 3 if db == "g3" else 9
    in1, in2 = (os.path.join(g1, "reads_a_1.fq"), os.path.join(g1, "reads_a_2.fq")) if db == "g3" else (os.path.join(d, "reads_1.fq"), os.path.join(d, "reads_2.fq"))
    got = None
    for k in range(n):
        p = O.default_params(L=151, insert_high=750, mode=1, top=95, nc=(k == 0))
        out = str(tmp_path / ("chain_%d.fq" % k))
        got = O.align_fastq_bam(os.path.join(d, "c%d.fa.index" % k), p, in1, in2, out, str(tmp_path / ("o%d" % k)), read_group="S.nt.tmp_in_1.fq")
        lines = open(out, "rb").read().split(b"\n")
        in1, in2 = out + ".1", out + ".2"
        with open(in1, "wb") as a, open(in2, "wb") as b:
            for i in range(0, len(lines) - 7, 8):
                a.write(b"\n".join(lines[i:i + 4]) + b"\n")
                b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")
    want = bam_golden(os.path.join(d, "expected_chain_a_b.bam.txt" if db == "g3" else "expected_chain_b.bam.txt"))
    from conftest import split_heap_garbage_trims
    odd = 0
    for k in ("dpout", "gout", "unpair"):
        w, g, n_odd = split_heap_garbage_trims(want[k], got[k])
        odd += n_odd
        assert g == w, k                                # the very order: the oracle's chain permutes the reads between runs as the reference's does
    assert odd == (2 if db == "g3" else 0)              # g3's chunks are single sequences without N runs: the reference's out-of-bounds read shows
    assert want["header"].startswith(got["header"])
    assert fastq_records(out) == fastq_records(os.path.join(d, "expected_chain_a.fq" if db == "g3" else "expected_chain.fq"))
