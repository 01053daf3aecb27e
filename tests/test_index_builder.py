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
    # .fmv: every explicit sample a rank query can reach (nearest-sample index <= (n+127)/256, BWT.c:614-615) and all
    # major counters must agree; the reference leaves the one sample past the text end undefined
    n = int(arrs["n"])
    a = np.frombuffer(open(prefix + ".fmv", "rb").read(), dtype=np.uint32)
    b = np.frombuffer(open(os.path.join(d, "ref.fa.index.fmv"), "rb").read(), dtype=np.uint32)
    assert len(a) == len(b) and np.array_equal(a[:10], b[:10])
    n_occ = (n + 255) // 256 + 1
    minor_words = (n_occ + 1) // 2 * 4
    last_used = (n + 127) // 256
    ma, mb = a[10:10 + minor_words].reshape(-1, 4), b[10:10 + minor_words].reshape(-1, 4)
    sa_ = np.stack([ma >> 16, ma & 0xFFFF], axis=1).reshape(-1, 4)[:last_used + 1]
    sb_ = np.stack([mb >> 16, mb & 0xFFFF], axis=1).reshape(-1, 4)[:last_used + 1]
    assert np.array_equal(sa_, sb_)
    assert np.array_equal(a[10 + minor_words:], b[10 + minor_words:])
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
