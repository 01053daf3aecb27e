"""-m gpu: the drop-in `soap4 pair ... -F` on the GPU must print exactly what the REAL reference printed for the same
command line (golden stdout from tests/golden/make_golden.py; order-insensitive)."""
import ctypes as C
import os
import subprocess

import pytest

import oracle_lib as O
from conftest import ROOT, fastq_records

pytestmark = pytest.mark.gpu


def _run_cli(args, out_path, driver="cxx", batch_pairs=None):
    """driver "cxx": bin/soap4-mi355x (C++ host code); "py": the Python mirror (megapath_amd/soap4.py)."""
    cmd = [os.path.join(ROOT, "bin", "soap4")] + args
    env = dict(os.environ)
    if driver == "py":
        env["MP_SOAP4_PYTHON"] = "1"
    else:
        assert os.path.exists(os.path.join(ROOT, "bin", "soap4-mi355x")), "C++ driver not built (make -C megapath_amd/csrc)"
        env.pop("MP_SOAP4_PYTHON", None)
    if batch_pairs:
        env["MP_BATCH_PAIRS"] = str(batch_pairs)
    with open(out_path, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()
    return r.stderr


@pytest.mark.parametrize("driver", ["cxx", "py"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_cli_stdout_equals_reference(golden_dir, tmp_path, tag, driver):
    d = os.path.join(golden_dir, "g1")
    L = int(open(os.path.join(d, "expected_%s.L" % tag)).read())
    out = tmp_path / "out.fq"
    err = _run_cli(["pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag),
                    "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "1", "-u", "750", "-F", "-nc"],
                   out, driver=driver, batch_pairs=70 if driver == "cxx" else None)   # several batches through the reader thread
    got = fastq_records(str(out))
    want = fastq_records(os.path.join(d, "expected_%s.fq" % tag))
    assert len(got) == len(want)
    assert got == want
    assert b"deep DP" in err


@pytest.mark.parametrize("driver", ["cxx", "py"])
def test_chained_chunks_merge_header_scores(golden_dir, tmp_path, driver):
    """NT stage chaining (runMegaPath.sh:191-227): the annotated FASTQ of one chunk is the input of the next; without
    -nc the SCORE: comments of the previous chunk are merged (BGS-IO.cpp:1348-1371).  Checked against the oracle."""
    d = os.path.join(golden_dir, "g1")
    lines = open(os.path.join(d, "expected_a.fq"), "rb").read().split(b"\n")
    f1, f2 = tmp_path / "c_1.fq", tmp_path / "c_2.fq"
    with open(f1, "wb") as a, open(f2, "wb") as b:          # what cc/deinterleave does between chunks
        for i in range(0, len(lines) - 7, 8):
            a.write(b"\n".join(lines[i:i + 4]) + b"\n")
            b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(d, "ref.fa.index"), str(f1), str(f2), "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
              "-L", "151", "-T", "1", "-u", "750", "-F", "-top", "95"], out, driver=driver)
    lib = O.oracle()
    ix = lib.mpo_index_open(os.path.join(d, "ref.fa.index").encode())
    p = O.default_params(L=151)
    st = O.AlignStats()
    ref_out = tmp_path / "oracle.fq"
    assert lib.mpo_align_fastq(ix, C.byref(p), str(f1).encode(), str(f2).encode(), str(ref_out).encode(), None, C.byref(st)) == 0
    assert fastq_records(str(out)) == fastq_records(str(ref_out))
    lib.mpo_index_close(ix)
