"""CPU tests: this repo's index builder (megapath_amd/index_builder.py) must write byte-identical .bwt/.fmv/.sa/.pac
to the files the REAL reference 2bwt-builder produced for the golden genome (tests/golden/g1)."""
import os

import numpy as np

from megapath_amd.index_builder import build_index_arrays, write_index_files


def _read_fasta(path):
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


def test_builder_matches_reference_files(golden_dir, tmp_path):
    d = os.path.join(golden_dir, "g1")
    arrs = build_index_arrays(_read_fasta(os.path.join(d, "ref.fa")), sa_interval=16)
    prefix = str(tmp_path / "mine.index")
    write_index_files(prefix, arrs, write_lkt=False)
    for ext in (".bwt", ".sa", ".pac"):
        a = open(prefix + ext, "rb").read()
        b = open(os.path.join(d, "ref.fa.index" + ext), "rb").read()
        assert a == b, ext
    # .fmv: byte-identical too, including the sample past the text end, which counts the cleared trailing code as 'A's
    # (BWTConstruct.c:312-314, 1036-1041) and is what rank queries in the last 128 positions decode backwards from
    assert open(prefix + ".fmv", "rb").read() == open(os.path.join(d, "ref.fa.index.fmv"), "rb").read()
    # annotation / translation carry the same coordinates (the reference also writes a random seed in .ann)
    mine = open(prefix + ".tra").read().split()
    ref = open(os.path.join(d, "ref.fa.index.tra")).read().split()
    assert mine == ref
    assert open(prefix + ".ann").read().split()[3:] == open(os.path.join(d, "ref.fa.index.ann")).read().split()[3:]


def test_oracle_loads_builder_output(golden_dir, tmp_path):
    import ctypes as C
    import oracle_lib as O
    d = os.path.join(golden_dir, "g1")
    arrs = build_index_arrays(_read_fasta(os.path.join(d, "ref.fa")), sa_interval=16)
    prefix = str(tmp_path / "mine.index")
    write_index_files(prefix, arrs)
    lib = O.oracle()
    a = lib.mpo_index_open(prefix.encode())
    b = lib.mpo_index_open(os.path.join(d, "ref.fa.index").encode())
    ta = np.ctypeslib.as_array(lib.mpo_lkt_table(a), shape=(4 ** 13,))
    tb = np.ctypeslib.as_array(lib.mpo_lkt_table(b), shape=(4 ** 13,))
    assert np.array_equal(ta, tb)
    for pos in (0, 1, 20000, 59999):
        pa, ca, pb, cb = C.c_uint64(), C.c_uint(), C.c_uint64(), C.c_uint()
        lib.mpo_chr_and_pos(a, pos, C.byref(pa), C.byref(ca))
        lib.mpo_chr_and_pos(b, pos, C.byref(pb), C.byref(cb))
        assert (pa.value, ca.value) == (pb.value, cb.value)
    lib.mpo_index_close(a)
    lib.mpo_index_close(b)


def test_rank_near_text_end_from_builder_files(tmp_path):
    """n mod 256 > 128: ranks in the tail decode backwards from the explicit sample past the text end (BWT.c:614-634).
    The oracle reading this builder's .fmv must agree with a direct count over the BWT for every index and symbol."""
    import oracle_lib as O
    from megapath_amd import synth
    genome = synth.random_genome(256 * 9 + 200, 77, n_seqs=2)
    arrs = build_index_arrays(genome, sa_interval=16)
    prefix = str(tmp_path / "tail.index")
    write_index_files(prefix, arrs)
    n = int(arrs["n"])
    bw = np.asarray(arrs["bwt_words"], dtype=np.uint32)
    syms = ((bw[:, None] >> (2 * (15 - np.arange(16, dtype=np.uint32)))) & 3).reshape(-1)[:n]
    inv = int(arrs["inv_sa0"])
    lib = O.oracle()
    ix = lib.mpo_index_open(prefix.encode())
    for c in range(4):
        cs = np.concatenate([[0], np.cumsum(syms == c)])
        for idx in list(range(n - 300, n + 2)) + [0, 1, inv, inv + 1]:
            k = idx - (1 if idx > inv else 0)           # the '$' row is not part of the stored BWT (BWT.c:605)
            assert lib.mpo_occ(ix, idx, c) == int(cs[k]), (idx, c)
    lib.mpo_index_close(ix)


def test_builder_removes_long_n_runs_like_the_reference(golden_dir, tmp_path):
    """tests/golden/nrun: a FASTA with runs of >= 10 ambiguity letters (removed and recorded in .tra, HSP.c:486-527), shorter runs
    (-> 'G'), IUPAC codes, lower case, 'U' (skipped), a gi| header, runs at the very start / end of a sequence -- and the index
    files the REAL 2bwt-builder wrote for it.  Everything byte-identical (.ann apart from the builder's random seed)."""
    from megapath_amd.index_builder import parse_fasta_reference_style
    d = os.path.join(golden_dir, "nrun")
    parsed = parse_fasta_reference_style(open(os.path.join(d, "n.fa"), "rb").read())
    assert len(parsed["ambiguity"]) == 5 and parsed["gi"][1] == 12345
    arrs = build_index_arrays(parsed, sa_interval=16)
    prefix = str(tmp_path / "mine.index")
    write_index_files(prefix, arrs, write_lkt=False)
    for ext in (".pac", ".bwt", ".fmv", ".sa", ".tra", ".amb"):
        assert open(prefix + ext, "rb").read() == open(os.path.join(d, "n.fa.index" + ext), "rb").read(), ext
    assert open(prefix + ".ann").read().split()[3:] == open(os.path.join(d, "n.fa.index.ann")).read().split()[3:]


def test_builder_reads_headers_like_the_reference(golden_dir, tmp_path):
    """tests/golden/hdr (files written by the REAL 2bwt-builder): a '>' inside a header line does not start a record (NT titles such as
    "3'->5' exonuclease"; HSP.c:441-449 reads the header to its newline), a header that begins with white space has an empty name, a
    last header without a newline is a record of length 0."""
    from megapath_amd.index_builder import parse_fasta_reference_style
    d = os.path.join(golden_dir, "hdr")
    parsed = parse_fasta_reference_style(open(os.path.join(d, "h.fa"), "rb").read())
    assert parsed["names"] == ["AB000001.1", "CD000002.2", "", "EF000003.1", "last_header_without_newline"]
    assert [len(s) for s in parsed["seqs"]] == [1500, 900, 700, 1100, 0]
    arrs = build_index_arrays(parsed, sa_interval=16)
    prefix = str(tmp_path / "mine.index")
    write_index_files(prefix, arrs, write_lkt=False)
    for ext in (".pac", ".bwt", ".fmv", ".sa", ".tra", ".amb"):
        assert open(prefix + ext, "rb").read() == open(os.path.join(d, "h.fa.index" + ext), "rb").read(), ext
    mine, ref = open(prefix + ".ann").read().split("\n"), open(os.path.join(d, "h.fa.index.ann")).read().split("\n")
    assert mine[1:] == ref[1:] and mine[0].split()[:2] == ref[0].split()[:2]
    # the advisor's example
    r = parse_fasta_reference_style(b">AB1.1 exo 3->5 domain\nACGTACGT\n>CD2.1 x\nGGGG\n")
    assert r["names"] == ["AB1.1", "CD2.1"] and [len(s) for s in r["seqs"]] == [8, 4]
