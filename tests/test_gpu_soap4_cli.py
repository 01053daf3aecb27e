"""-m gpu: the drop-in `soap4 pair ... -F` on the GPU must print exactly what the REAL reference printed for the same
command line (golden stdout from tests/golden/make_golden.py; order-insensitive)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
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


@pytest.mark.parametrize("driver", ["cxx", "py"])
def test_paired_only_mode_gz_input_and_ignore_comments(golden_dir, tmp_path, driver):
    """-P ("not paired == not aligned", BGS-IO.cpp:1393-1396, 2003-2011), gzip'ed inputs (kseq/zlib in the reference), and
    reads carrying the IGNORE comment (BGS-IO.cpp:1400-1402, 2046-2048).  Expected output: the oracle in mode 2."""
    import gzip
    d = os.path.join(golden_dir, "g1")
    src1 = open(os.path.join(d, "reads_a_1.fq"), "rb").read().split(b"\n")
    src2 = open(os.path.join(d, "reads_a_2.fq"), "rb").read().split(b"\n")
    # every 7th pair is marked IGNORE on read 1, every 11th on both mates
    for i in range(0, len(src1) - 3, 4):
        k = i // 4
        if k % 7 == 0:
            src1[i] = src1[i] + b" IGNORE"
        if k % 11 == 0:
            src1[i] = src1[i].split(b" ")[0] + b" IGNORE"
            src2[i] = src2[i] + b" IGNORE"
    f1, f2 = tmp_path / "p_1.fq", tmp_path / "p_2.fq"
    f1.write_bytes(b"\n".join(src1))
    f2.write_bytes(b"\n".join(src2))
    g1, g2 = tmp_path / "p_1.fq.gz", tmp_path / "p_2.fq.gz"
    for a, b in ((f1, g1), (f2, g2)):
        with gzip.open(b, "wb") as z:
            z.write(a.read_bytes())
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(d, "ref.fa.index"), str(g1), str(g2), "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
              "-L", "151", "-T", "1", "-u", "750", "-P"], out, driver=driver)
    lib = O.oracle()
    ix = lib.mpo_index_open(os.path.join(d, "ref.fa.index").encode())
    p = O.default_params(L=151, mode=2)
    st = O.AlignStats()
    ref_out = tmp_path / "oracle.fq"
    assert lib.mpo_align_fastq(ix, C.byref(p), str(f1).encode(), str(f2).encode(), str(ref_out).encode(), None, C.byref(st)) == 0
    lib.mpo_index_close(ix)
    got, want = fastq_records(str(out)), fastq_records(str(ref_out))
    assert got == want
    assert any(b"IGNORE" in h.encode() for _, _, h, _, _ in got)


@pytest.mark.parametrize("driver", ["cxx", "py"])
@pytest.mark.parametrize("tag,reads,flags", [("d", "d", ["-P"]), ("f", "d", ["-P", "-nc"]), ("e", "e", ["-F", "-top", "95"])])
def test_cli_equals_reference_paired_only_and_chained(golden_dir, tmp_path, tag, reads, flags, driver):
    d = os.path.join(golden_dir, "g1")
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_%s_1.fq" % reads), os.path.join(d, "reads_%s_2.fq" % reads),
              "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "1", "-u", "750"] + flags, out, driver=driver)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_%s.fq" % tag))


# (set, -L, mode flag, extra flags): the commands the BAM goldens were made with (tests/golden/make_golden_bam.py)
# "<set>.bonly": `-b` alone -- nothing on stdout, BAM records made from the results as the DP left them (BGS-IO.cpp:2103: the
# MegaPath modes rewrite boundary-spanning mates and the pair scores before the records are made; set h is all boundary pairs)
BAM_RUNS = {"a": (151, "-F", ["-nc"]), "c": (301, "-F", ["-nc"]), "d": (151, "-P", []), "g": (151, "-F", ["-p"]), "h": (151, "-F", []),
            "a.bonly": (151, None, []), "g.bonly": (151, None, ["-p"]), "h.bonly": (151, None, [])}


@pytest.mark.parametrize("tag", sorted(BAM_RUNS))
def test_cli_bam_files_equal_reference(golden_dir, tmp_path, tag):
    """`soap4 ... -b` (row a15): the three kinds of BAM side files, decoded record by record, against what the REAL
    reference wrote for the same command (-T 1): flag, position, MAPQ, CIGAR, mate fields, tlen, sequence, qualities and
    the RG X0 X1 XM XO XG [NM MD] XA PH tags, in the same order; stdout must stay byte-identical too."""
    import bamtext
    from conftest import bam_golden
    d = os.path.join(golden_dir, "g1")
    L, mode, extra = BAM_RUNS[tag]
    out = tmp_path / "out.fq"
    prefix = str(tmp_path / "bam")
    rk = tag.split(".")[0]
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", "ref.fa.index", "reads_%s_1.fq" % rk, "reads_%s_2.fq" % rk, "-o", prefix,
           "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "2", "-u", "750", "-b"] + ([mode] if mode else []) + extra
    env = dict(os.environ, MP_BATCH_PAIRS="1048576")
    with open(out, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, cwd=d, env=env)     # relative names: RG = "reads_<set>_1.fq" as in the golden
    assert r.returncode == 0, r.stderr.decode()
    want = bam_golden(os.path.join(d, "expected_%s.bam.txt" % tag))
    text, refs, dp = bamtext.read_bam(prefix + ".dpout.1")
    gout = bamtext.read_bam(prefix + ".gout.1")[2]
    assert bamtext.read_bam(prefix + ".gout.2")[2] == []          # one header-only file per further -T thread
    unpair = bamtext.read_bam(prefix + ".unpair")[2]
    assert dp == want["dpout"]
    assert gout == want["gout"]
    assert unpair == want["unpair"]
    sq = [l for l in want["header"].splitlines() if not l.startswith("@PG")]
    assert [l for l in text.splitlines() if not l.startswith("@PG")] == sq
    assert [("@SQ\tSN:%s\tLN:%d" % r) for r in refs] == [l for l in sq if l.startswith("@SQ")]
    assert open(out, "rb").read() == (open(os.path.join(d, "expected_%s.fq" % tag), "rb").read() if mode else b"")


def test_cli_without_an_output_mode_is_refused(golden_dir, tmp_path):
    """`<aligner> pair <index> <r1> <r2> -o <prefix>` is what runSOAP3dp.pl:44-50 starts: no -F / -P / -b.  The reference dies with
    SIGSEGV after seeding on that command line (seen again with oracle/_ref/soap4 when the `.bonly` goldens were
    made); the drop-in refuses it up front: exit code 1, a message that names the missing option, nothing written."""
    d = os.path.join(golden_dir, "g1")
    prefix = str(tmp_path / "o")
    r = subprocess.run([os.path.join(ROOT, "bin", "soap4"), "pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"),
                        "-o", prefix, "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "1", "-u", "750"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
    assert r.stdout == b""
    assert b"-F" in r.stderr and b"-b" in r.stderr
    assert not [f for f in os.listdir(tmp_path) if f.startswith("o.")]


@pytest.mark.parametrize("tag,L,mode,extra,reads", [("a", 151, "-F", ["-nc", "-lsam"], "a"), ("a.noseq", 151, "-F", ["-nc", "-lsam-noseq"], "a"),
                                                    ("d", 151, "-P", ["-lsam"], "d"), ("e", 151, "-F", ["-top", "95", "-lsam"], "e"),
                                                    ("g", 151, "-F", ["-lsam"], "g"), ("h", 151, "-F", ["-lsam"], "h")])
def test_cli_direct_lsam_equals_reference_pipe(golden_dir, tmp_path, tag, L, mode, extra, reads):
    """-lsam (extension): stdout is what `soap4 ... | cc/fastq2lsam 1` gives in the reference pipeline (runMegaPath.sh:136,
    162,208) -- byte-identical to the real tool run on the real soap4's output, chained-chunk comments and IGNORE included."""
    d = os.path.join(golden_dir, "g1")
    out = tmp_path / "out.lsam"
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_%s_1.fq" % reads),
           os.path.join(d, "reads_%s_2.fq" % reads), "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L),
           "-T", "3", "-u", "750", mode] + extra
    with open(out, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert open(out, "rb").read() == open(os.path.join(d, "expected_%s.lsam" % tag), "rb").read()


def _oracle_stdout(prefix, f1, f2, L, tmp_path, mode=1, nc=True):
    lib = O.oracle()
    ix = lib.mpo_index_open(prefix.encode())
    p = O.default_params(L=L, mode=mode, nc=nc)
    st = O.AlignStats()
    out = str(tmp_path / "oracle_edge.fq")
    assert lib.mpo_align_fastq(ix, C.byref(p), f1.encode(), f2.encode(), out.encode(), None, C.byref(st)) == 0
    lib.mpo_index_close(ix)
    return out


def test_cli_edge_cases_empty_single_ragged_and_mismatched_inputs(golden_dir, tmp_path):
    """Empty read files, a single pair, a batch boundary in the middle of the input, FASTA input, and files with
    different numbers of records (the reference exits with 'number of sequences of pair-end files not matched')."""
    d = os.path.join(golden_dir, "g1")
    prefix = os.path.join(d, "ref.fa.index")
    ini = os.path.join(ROOT, "tests", "soap4-nt2.ini")
    base = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefix]
    tail = ["-o", str(tmp_path / "o"), "-C", ini, "-L", "151", "-T", "2", "-u", "750", "-F", "-nc"]
    e1, e2 = tmp_path / "e_1.fq", tmp_path / "e_2.fq"
    e1.write_bytes(b""); e2.write_bytes(b"")
    r = subprocess.run(base + [str(e1), str(e2)] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()
    src1 = open(os.path.join(d, "reads_a_1.fq"), "rb").read().split(b"\n")
    src2 = open(os.path.join(d, "reads_a_2.fq"), "rb").read().split(b"\n")
    for n_pairs, batch in ((1, None), (7, 3)):
        f1, f2 = tmp_path / ("s%d_1.fq" % n_pairs), tmp_path / ("s%d_2.fq" % n_pairs)
        f1.write_bytes(b"\n".join(src1[:4 * n_pairs]) + b"\n"); f2.write_bytes(b"\n".join(src2[:4 * n_pairs]) + b"\n")
        env = dict(os.environ)
        if batch:
            env["MP_BATCH_PAIRS"] = str(batch)
        r = subprocess.run(base + [str(f1), str(f2)] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        got = tmp_path / "got_edge.fq"
        got.write_bytes(r.stdout)
        assert fastq_records(str(got)) == fastq_records(_oracle_stdout(prefix, str(f1), str(f2), 151, tmp_path))
    # FASTA input (kseq reads both): no quality line in, empty quality line out
    fa1, fa2 = tmp_path / "fa_1.fa", tmp_path / "fa_2.fa"
    fa1.write_bytes(b"".join(b">" + src1[i][1:] + b"\n" + src1[i + 1] + b"\n" for i in range(0, 40, 4)))
    fa2.write_bytes(b"".join(b">" + src2[i][1:] + b"\n" + src2[i + 1] + b"\n" for i in range(0, 40, 4)))
    r = subprocess.run(base + [str(fa1), str(fa2)] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.split(b"\n")
    assert len(lines) == 10 * 2 * 4 + 1 and all(lines[i + 3] == b"" for i in range(0, 80, 4))
    # ragged: one file has a record more
    g1 = tmp_path / "g_1.fq"
    g1.write_bytes(b"\n".join(src1[:4 * 5]) + b"\n")
    g2 = tmp_path / "g_2.fq"
    g2.write_bytes(b"\n".join(src2[:4 * 4]) + b"\n")
    r = subprocess.run(base + [str(g1), str(g2)] + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"number of sequences of pair-end files not matched" in r.stderr


def test_cli_longest_supported_reads(tmp_path):
    """Reads of 330-384 bases (the 32-lane x 12-column DP shape, 444-row windows, 751-row rescue windows) against the
    oracle; one base more than the kernels cover is refused with a message instead of computed wrongly."""
    from megapath_amd import index_builder, synth
    genome = synth.random_genome(300000, 91, n_seqs=2, repeat_frac=0.05)
    prefix = str(tmp_path / "long.index")
    index_builder.write_index_files(prefix, index_builder.build_index_arrays(genome, sa_interval=16))
    names, r1, r2 = synth.simulate_pairs(genome, 300, 384, 92, sub=0.02, ins=0.002, dele=0.002, random_frac=0.05, broken_mate_frac=0.1,
                                         noisy_mate_frac=0.2, noisy_sub=0.15, len_range=(330, 384), frag_mean=600, frag_sd=40, frag_min=400, frag_max=740)
    f1, f2 = str(tmp_path / "l_1.fq"), str(tmp_path / "l_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefix, f1, f2, "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
           "-T", "2", "-u", "750", "-F", "-nc"]
    r = subprocess.run(cmd + ["-L", "385"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    got = tmp_path / "got_long.fq"
    got.write_bytes(r.stdout)
    assert fastq_records(str(got)) == fastq_records(_oracle_stdout(prefix, f1, f2, 385, tmp_path))
    # 385-base reads: refused
    f3, f4 = str(tmp_path / "x_1.fq"), str(tmp_path / "x_2.fq")
    open(f3, "wb").write(b"@x/1\n" + b"ACGT" * 97 + b"\n+\n" + b"I" * 388 + b"\n")
    open(f4, "wb").write(b"@x/2\n" + b"TGCA" * 97 + b"\n+\n" + b"I" * 388 + b"\n")
    r = subprocess.run(cmd[:3] + [f3, f4] + cmd[5:] + ["-L", "389"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"mp_" in r.stderr


@pytest.mark.parametrize("driver", ["cxx", "py"])
@pytest.mark.parametrize("tag,L", [("t", 301), ("s", 151)])
def test_cli_g2_tied_candidates_and_unseedable_reads(golden_dir, tmp_path, tag, L, driver):
    """g2/t: reads with > 200 tied single-end candidates -- the survivors of the cut at 200 depend on the tie order of the
    reference's std::sort (DV-DPfunctions.cpp:334, DV-DPForSingleReads.cpp:200); g2/s: mates of 1..16 bases.  Both are the
    real reference's stdout."""
    d = os.path.join(golden_dir, "g2")
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_%s_1.fq" % tag), os.path.join(d, "reads_%s_2.fq" % tag),
              "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "2", "-u", "750", "-F", "-nc"],
             out, driver=driver)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_%s.fq" % tag))


def test_cli_reads_below_the_dp_cutoff_do_not_abort_or_hurt_neighbours(golden_dir, tmp_path):
    """Reads of 17..29 bases can be seeded but never reach the cutoff floor of 30.  The reference gives up on the whole SIMD
    group such a DP instance shares (CPU_DP.cpp:294-299): not reproduced.  Here they come out unaligned, every other read
    as the oracle aligns it, also with -b."""
    from megapath_amd import synth
    d = os.path.join(golden_dir, "g1")
    prefix = os.path.join(d, "ref.fa.index")
    genome = []
    name, seq = None, []
    for line in open(os.path.join(d, "ref.fa"), "rb").read().split(b"\n"):
        if line.startswith(b">"):
            if name:
                genome.append((name, np.frombuffer(b"".join(seq), dtype=np.uint8)))
            name, seq = line[1:].decode(), []
        elif line:
            seq.append(line)
    genome.append((name, np.frombuffer(b"".join(seq), dtype=np.uint8)))
    names, r1, r2 = synth.simulate_pairs(genome, 300, 150, 4242, sub=0.01, len_range=(30, 150), broken_mate_frac=0.1)
    rng = np.random.default_rng(5)
    for k in range(len(names)):
        if k % 3 == 0:
            r1[k] = r1[k][:int(rng.integers(17, 30))]
        if k % 4 == 0:
            r2[k] = r2[k][:int(rng.integers(1, 30))]
    f1, f2 = str(tmp_path / "w_1.fq"), str(tmp_path / "w_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    for extra in ([], ["-b"]):
        cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefix, f1, f2, "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
               "-L", "151", "-T", "2", "-u", "750", "-F", "-nc"] + extra
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        got = tmp_path / "got_short.fq"
        got.write_bytes(r.stdout)
        recs = fastq_records(str(got))
        assert recs == fastq_records(_oracle_stdout(prefix, f1, f2, 151, tmp_path))
        assert all("SCORE:0;" in x[2] for x in recs if len(x[3]) < 30)


@pytest.mark.parametrize("tag,L", [("a", 151), ("c", 301)])
def test_cli_resident_chunks_equal_reference_chain(golden_dir, tmp_path, tag, L):
    """Row f2: three index chunks resident in one process, every batch aligned against each, per-reference hits merged as the
    chained reference runs merge them through the SCORE: comments -- equals the last stdout of the real three-run chain (g3)."""
    g1, g3 = os.path.join(golden_dir, "g1"), os.path.join(golden_dir, "g3")
    out = tmp_path / "out.fq"
    prefixes = ",".join(os.path.join(g3, "c%d.fa.index" % k) for k in range(3))
    err = _run_cli(["pair", prefixes, os.path.join(g1, "reads_%s_1.fq" % tag), os.path.join(g1, "reads_%s_2.fq" % tag), "-o", str(tmp_path / "o"),
                    "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "2", "-u", "750", "-F", "-nc", "-top", "95"], out, batch_pairs=50)
    assert fastq_records(str(out)) == fastq_records(os.path.join(g3, "expected_chain_%s.fq" % tag))
    assert b"3 index chunk(s)" in err


@pytest.mark.parametrize("tag,L", [("a", 151), ("c", 301)])
def test_cli_host_merge_path_equals_reference(golden_dir, tmp_path, tag, L):
    """MP_HOST_MERGE=1: the hit lists travel to the host and the header comments are merged there chunk after chunk (the path -b always takes);
    the default without -b reduces the per-reference scores on the device (mp_reduce_*).  Both must print the real soap4's records: one chunk
    (g1) and the three chained chunks of g3."""
    g1, g3 = os.path.join(golden_dir, "g1"), os.path.join(golden_dir, "g3")
    for prefixes, want, extra in ((os.path.join(g1, "ref.fa.index"), os.path.join(g1, "expected_%s.fq" % tag), []),
                                  (",".join(os.path.join(g3, "c%d.fa.index" % k) for k in range(3)), os.path.join(g3, "expected_chain_%s.fq" % tag), ["-top", "95"])):
        outs = []
        for host in (True, False):
            out = tmp_path / ("out_%d.fq" % host)
            cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefixes, os.path.join(g1, "reads_%s_1.fq" % tag), os.path.join(g1, "reads_%s_2.fq" % tag), "-o", str(tmp_path / "o"),
                   "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "2", "-u", "750", "-F", "-nc"] + extra
            env = dict(os.environ, MP_BATCH_PAIRS="50")
            if host:
                env["MP_HOST_MERGE"] = "1"
            with open(out, "wb") as fo:
                r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 0, r.stderr.decode()
            assert (b"reduced on the device" in r.stderr) == (not host)
            outs.append(fastq_records(str(out)))
        assert outs[0] == outs[1] == fastq_records(want)


@pytest.mark.parametrize("rewrite", [False, True, "device", "lean", "lean_plain_stash", "lean_by_itself"])
def test_cli_nine_resident_chunks_equal_reference_chain(golden_dir, tmp_path, rewrite):
    """g6: nine index chunks resident in one process (homologous sequences at 0-7 % divergence in several of them) against the last
    stdout of NINE chained runs of the real soap4; with the untouched comments left alone from the second chunk on (the default) and
    with every comment merged again in every chunk (MP_CARRY_REWRITE=1)."""
    g6 = os.path.join(golden_dir, "g6")
    out = tmp_path / "out.fq"
    prefixes = ",".join(os.path.join(g6, "c%d.fa.index" % k) for k in range(9))
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefixes, os.path.join(g6, "reads_1.fq"), os.path.join(g6, "reads_2.fq"), "-o", str(tmp_path / "o"),
           "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "3", "-u", "750", "-F", "-nc", "-top", "95"]
    env = dict(os.environ, MP_BATCH_PAIRS="64")
    if rewrite == "device":
        pass                                             # the default: per-reference scores reduced on the device over the nine chunks
    elif str(rewrite).startswith("lean"):
        # the NT stage's layout: no optional table, the 13-mer table as 8-byte entries with a seed's first extension step in them (MP_LKT14); the seed
        # positions between the two passes of the -top rule as 8-byte words (default) or as they stand
        env.update(MP_KMER_FILTER="0", MP_KMER15="0", MP_LKT16="0", MP_SA_DENSE="0", MP_LKT14="1")
        if rewrite == "lean_plain_stash":
            env["MP_STASH_PLAIN"] = "1"
        if rewrite == "lean_by_itself":
            # no switch given: the executable decides the layout for the database as a whole (here as if the card were too small for the optional tables)
            for k in ("MP_KMER_FILTER", "MP_KMER15", "MP_LKT16", "MP_SA_DENSE", "MP_LKT14"):
                env.pop(k, None)
            env["MP_LAYOUT_HBM_BYTES"] = "1000000"
    else:
        env["MP_HOST_MERGE"] = "1"                       # hit lists to the host, comments merged chunk after chunk
        if rewrite:
            env["MP_CARRY_REWRITE"] = "1"
    with open(out, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert b"9 index chunk(s)" in r.stderr and (b"reduced on the device" in r.stderr) == (rewrite == "device" or str(rewrite).startswith("lean"))
    assert (b"lean layout for all of them" in r.stderr) == (rewrite == "lean_by_itself")
    assert fastq_records(str(out)) == fastq_records(os.path.join(g6, "expected_chain.fq"))


@pytest.mark.parametrize("path", ["device_reduce", "host_merge"])
@pytest.mark.parametrize("db", ["g6", "g3"])
def test_cli_bam_with_resident_chunks_equals_reference_chain(golden_dir, tmp_path, db, path):
    """The NT stage's real command line has -b (runMegaPath.sh:199) and every run of the chain writes the same -o prefix: what remains are the
    LAST chunk's BAM files and the last run's stdout (goldens from the real chain, tests/golden/make_golden_g7.py).  With the chunks resident
    the comments come from the device reduction over all chunks and only the last chunk's hit lists travel to the host for the BAM records
    (default); MP_BAM_HOST_MERGE=1 is the round-3 path (every chunk's lists to the host).  Both: stdout and every BAM record equal."""
    import shutil
    import bamtext
    from conftest import bam_golden
    g1 = os.path.join(golden_dir, "g1")
    d = os.path.join(golden_dir, db)
    n = 9 if db == "g6" else 3
    reads = (os.path.join(d, "reads_1.fq"), os.path.join(d, "reads_2.fq")) if db == "g6" else (os.path.join(g1, "reads_a_1.fq"), os.path.join(g1, "reads_a_2.fq"))
    want_fq = os.path.join(d, "expected_chain.fq" if db == "g6" else "expected_chain_a.fq")
    want = bam_golden(os.path.join(d, "expected_chain_b.bam.txt" if db == "g6" else "expected_chain_a_b.bam.txt"))
    # the chain's runs read <prefix>.nt.tmp_in_1.fq; the read group is the first read file as given (SOAP4.cpp:297-300)
    for m in (1, 2):
        shutil.copy(reads[m - 1], tmp_path / ("S.nt.tmp_in_%d.fq" % m))
    prefixes = ",".join(os.path.join(d, "c%d.fa.index" % k) for k in range(n))
    out = tmp_path / "out.fq"
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefixes, "S.nt.tmp_in_1.fq", "S.nt.tmp_in_2.fq", "-b", "-o", "S.nt.bam", "-L", "151", "-T", "1", "-u", "750", "-F",
           "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-top", "95", "-nc"]
    env = dict(os.environ, MP_BATCH_PAIRS="64")
    if path == "host_merge":
        env["MP_BAM_HOST_MERGE"] = "1"
    with open(out, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    assert (b"reduced on the device" in r.stderr) == (path == "device_reduce")
    assert fastq_records(str(out)) == fastq_records(want_fq)
    prefix = str(tmp_path / "S.nt.bam")
    text, refs, dp = bamtext.read_bam(prefix + ".dpout.1")
    gout = bamtext.read_bam(prefix + ".gout.1")[2]
    unpair = bamtext.read_bam(prefix + ".unpair")[2]
    # batches of 64 pairs: the sections of the files are per batch; the reference made one batch of everything -- same records, compared as sorted lists per file
    from conftest import split_heap_garbage_trims

    def same(got_recs, want_recs):
        w, g, _ = split_heap_garbage_trims(want_recs, got_recs)     # (g3: two records the reference damages through an out-of-bounds read)
        return sorted(g) == sorted(w)
    assert same(dp, want["dpout"]) and same(gout, want["gout"]) and same(unpair, want["unpair"])
    sq = [l for l in want["header"].splitlines() if not l.startswith("@PG")]
    assert [l for l in text.splitlines() if not l.startswith("@PG")] == sq
    # (the ORDER of the records is the chain's own business: every run prints its deep-DP pairs first, then the rescued ones, then the rest, and the
    # next run reads that as its input order -- the last run of the chain sees the reads permuted by the eight before it; SURVEY App. D12: order is
    # not part of the contract, it also changes with -T)
    env["MP_BATCH_PAIRS"] = "1048576"                  # one batch
    with open(out, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    assert same(bamtext.read_bam(prefix + ".dpout.1")[2], want["dpout"]) and same(bamtext.read_bam(prefix + ".gout.1")[2], want["gout"])
    assert same(bamtext.read_bam(prefix + ".unpair")[2], want["unpair"])
    assert fastq_records(str(out)) == fastq_records(want_fq)


@pytest.mark.parametrize("mode", ["-F", "-P"])
def test_cli_many_resident_chunks_leave_untouched_comments_alone(golden_dir, tmp_path, mode):
    """With many chunks most reads hit few of them: from the second chunk on, a read without a hit in a chunk keeps the comment the
    chunk before left (it is a fixed point of the merge).  Seven resident chunks, against the same run with MP_CARRY_REWRITE=1
    (every comment merged again in every chunk) -- and the three-chunk prefix of it is pinned on the real chain by the test above."""
    g1, g3 = os.path.join(golden_dir, "g1"), os.path.join(golden_dir, "g3")
    order = [0, 1, 2, 1, 0, 2, 1]
    prefixes = ",".join(os.path.join(g3, "c%d.fa.index" % k) for k in order)
    outs = []
    for rewrite in (False, True):
        out = tmp_path / ("out_%d.fq" % rewrite)
        cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefixes, os.path.join(g1, "reads_a_1.fq"), os.path.join(g1, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
               "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "3", "-u", "750", mode, "-nc", "-top", "95"]
        env = dict(os.environ, MP_BATCH_PAIRS="60", MP_HOST_MERGE="1")
        if rewrite:
            env["MP_CARRY_REWRITE"] = "1"
        with open(out, "wb") as fo:
            r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        assert b"7 index chunk(s)" in r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 100000


def test_cli_same_chunk_twice_equals_chained_golden_e(golden_dir, tmp_path):
    """The chained two-run golden of g1 (set e: run 2 reads run 1's stdout with its comments) from ONE process holding the index twice."""
    d = os.path.join(golden_dir, "g1")
    out = tmp_path / "out.fq"
    prefix = os.path.join(d, "ref.fa.index")
    _run_cli(["pair", prefix + "," + prefix, os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
              "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "2", "-u", "750", "-F", "-nc", "-top", "95"], out)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_e.fq"))


def test_cli_two_workers_keep_input_order_of_batches(golden_dir, tmp_path):
    """Several workers finish batches out of order; the printer writes them in input order: with one pair per read name the
    sequence of batches in the output is the input's."""
    d = os.path.join(golden_dir, "g1")
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
              "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "2", "-u", "750", "-F", "-nc"], out, batch_pairs=16)
    names = [l.split(b"\t")[0][1:] for l in open(out, "rb") if l.startswith(b"@r")][0::2]
    src = [l.split()[0][1:-2] for l in open(os.path.join(d, "reads_a_1.fq"), "rb") if l.startswith(b"@r")]
    batch_of = {n: i // 16 for i, n in enumerate(src)}
    seq = [batch_of[n] for n in names]
    assert seq == sorted(seq) and len(names) == len(src)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_a.fq"))


def test_cli_lsam_id_and_pass1_counters_equal_reference_tools(golden_dir, tmp_path):
    """Row f3: -lsamid prints what `soap4 | cc/fastq2lsam 1 | cc/taxLookupAcc` prints (species-level taxids, super-kingdom names),
    -pass1 leaves the counters of cc/reassign's pass 1 over those lines; tests/golden/g4 holds the real tools' outputs."""
    from megapath_amd import taxcount
    g4 = os.path.join(golden_dir, "g4")
    out = tmp_path / "out.lsam.id"
    counters = tmp_path / "pass1.txt"
    _run_cli(["pair", os.path.join(g4, "ref.fa.index"), os.path.join(g4, "reads_1.fq"), os.path.join(g4, "reads_2.fq"), "-o", str(tmp_path / "o"),
              "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "3", "-u", "750", "-F", "-nc", "-top", "95",
              "-lsamid", os.path.join(g4, "acc2tid.txt"), os.path.join(g4, "nodes.dmp"), os.path.join(g4, "names.dmp"), "-pass1", str(counters)], out, batch_pairs=100)
    want = open(os.path.join(g4, "expected.lsam.id"), "rb").read().splitlines()
    got = open(out, "rb").read().splitlines()
    assert sorted(got) == sorted(want)
    assert any(b";" in l.split(b"\t")[5] for l in want) and any(l.endswith(b"\tViruses") for l in want)
    p = taxcount.read_pass1_file(str(counters))
    assert p == taxcount.pass1_from_lsam_id([l + b"\n" for l in want], 40)
    assert sorted("%d explains %d" % e for e in taxcount.explains(p)) == sorted(l for l in open(os.path.join(g4, "expected.explains")).read().split("\n") if l)


@pytest.mark.parametrize("k,overlap", [(1, 0.5), (2, 0.3)])
def test_cli_spike_filter_equals_the_file_pipeline(golden_dir, tmp_path, k, overlap):
    """Row f4: -spike k overlap (with -b) leaves <prefix>.genomecovFilter and <prefix>.readFilter computed on the device -- what
    samtools merge *[0-9] | bedtools genomecov -bga | cc/genomeCovFilter k | bedtools bamtobed + annotate | awk would leave
    (runMegaPath.sh:211-218).  Checker: tests/spike_check.py on the BAM files the same run wrote (record-identical to the
    reference's, see test_cli_bam_files_equal_reference); the mean + k*stdev step runs the REAL cc/genomeCovFilter when its binary
    is here, the bedtools steps are restated (third-party binary, absent)."""
    import bamtext
    import spike_check as S
    d = os.path.join(golden_dir, "g1")
    prefix = str(tmp_path / "bam")
    # a pile-up so that spikes exist: every 10th pair of set "a" repeated eight times
    src1 = open(os.path.join(d, "reads_a_1.fq"), "rb").read().split(b"\n")
    src2 = open(os.path.join(d, "reads_a_2.fq"), "rb").read().split(b"\n")
    f1, f2 = tmp_path / "p_1.fq", tmp_path / "p_2.fq"
    with open(f1, "wb") as a, open(f2, "wb") as b:
        for i in range(0, len(src1) - 3, 4):
            reps = 8 if (i // 4) % 10 == 0 else 1
            for r in range(reps):
                tag = b"" if r == 0 else b"x%d" % r
                a.write(src1[i].replace(b"/1", tag + b"/1") + b"\n" + b"\n".join(src1[i + 1:i + 4]) + b"\n")
                b.write(src2[i].replace(b"/2", tag + b"/2") + b"\n" + b"\n".join(src2[i + 1:i + 4]) + b"\n")
    cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", os.path.join(d, "ref.fa.index"), str(f1), str(f2), "-o", prefix, "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
           "-L", "151", "-T", "2", "-u", "750", "-F", "-nc", "-b", "-spike", str(k), str(overlap)]
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MP_BATCH_PAIRS="97"))
    assert r.returncode == 0, r.stderr.decode()
    _, refs, dp = bamtext.read_bam(prefix + ".dpout.1")
    recs = dp + bamtext.read_bam(prefix + ".gout.1")[2] + bamtext.read_bam(prefix + ".gout.2")[2]
    bed = S.bed_from_bam_text(recs)
    genome = [(n, ln) for n, ln in refs]
    bg = S.bedgraph_bga(genome, bed)
    mine, _ = S.genome_cov_filter(genome, bg, k)
    if os.path.exists(S.REAL):
        assert S.real_genome_cov_filter(genome, bg, k, str(tmp_path)) == mine
    want_iv = S.union(mine)
    got_iv = [(l.split("\t")[0], int(l.split("\t")[1]), int(l.split("\t")[2])) for l in open(prefix + ".genomecovFilter").read().splitlines()]
    assert got_iv == want_iv and len(want_iv) > 0
    want_names = sorted(S.read_filter(bed, mine, overlap))
    got_names = sorted(open(prefix + ".readFilter").read().split())
    assert got_names == want_names and 0 < len(want_names) < len(bed)


def test_builder_cli_then_align(golden_dir, tmp_path):
    """`bin/2bwt-builder ref.fa` (the reference builder's command line) on the golden genome: the files equal the real builder's, and
    `bin/soap4` aligning against the freshly built index prints the golden stdout of the real soap4."""
    import shutil
    d = os.path.join(golden_dir, "g1")
    fa = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(d, "ref.fa"), fa)
    r = subprocess.run([os.path.join(ROOT, "bin", "2bwt-builder"), fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    for ext in (".bwt", ".fmv", ".sa", ".pac", ".tra", ".amb"):
        assert open(fa + ".index" + ext, "rb").read() == open(os.path.join(d, "ref.fa.index" + ext), "rb").read(), ext
    assert os.path.getsize(fa + ".index.lkt") == 4 + 8 * 4 ** 13
    out = tmp_path / "out.fq"
    _run_cli(["pair", fa + ".index", os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
              "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "2", "-u", "750", "-F", "-nc"], out)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_a.fq"))
    r = subprocess.run([os.path.join(ROOT, "bin", "2bwt-builder"), str(tmp_path / "missing.fa")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Cannot open" in r.stderr


SOAK = [  # (seed, genome bases, sequences, repeat fraction, pairs, read length, -L, simulate_pairs keywords)
    (11, 400000, 3, 0.15, 2500, 150, 151, dict(sub=0.01, ins=0.001, dele=0.001, random_frac=0.05)),
    (12, 250000, 1, 0.30, 2000, 150, 151, dict(sub=0.03, ins=0.003, dele=0.003, random_frac=0.10, broken_mate_frac=0.1, junk_end_frac=0.2, junk_end_max=40)),
    (13, 300000, 5, 0.05, 2000, 100, 101, dict(sub=0.02, random_frac=0.05, len_range=(40, 100), noisy_mate_frac=0.2, noisy_sub=0.12)),
    (14, 500000, 2, 0.10, 1500, 250, 251, dict(sub=0.015, ins=0.002, dele=0.002, random_frac=0.05, long_indel_frac=0.2, long_indel_max=20,
                                               frag_mean=450, frag_sd=50, frag_min=300, frag_max=700)),
    (15, 200000, 2, 0.50, 1500, 150, 151, dict(sub=0.005, random_frac=0.3, n_frac=0.002)),
    (16, 350000, 4, 0.10, 1500, 300, 301, dict(sub=0.02, ins=0.001, dele=0.001, random_frac=0.1, broken_mate_frac=0.3, len_range=(75, 300),
                                               frag_mean=500, frag_sd=60, frag_min=320, frag_max=740)),
]


@pytest.mark.parametrize("cfg", SOAK, ids=[str(c[0]) for c in SOAK])
def test_cli_equals_oracle_on_random_workloads(tmp_path, cfg):
    """Whole path through the executable against the oracle's whole path on freshly generated genomes and read sets of several shapes
    (repeat-rich, divergent, short / long / mixed lengths, junk ends, long indels, broken and noisy mates, many random pairs): the
    index is built by bin/2bwt-builder's code path, every optional table is on (the index is small), two batches per worker."""
    from megapath_amd import index_builder, synth
    seed, n_bases, n_seqs, rep, n_pairs, read_len, L, kw = cfg
    genome = synth.random_genome(n_bases, 7000 + seed, n_seqs=n_seqs, repeat_frac=rep)
    prefix = str(tmp_path / "soak.index")
    index_builder.write_index_files(prefix, index_builder.build_index_arrays(genome, sa_interval=16))
    names, r1, r2 = synth.simulate_pairs(genome, n_pairs, read_len, 8000 + seed, **kw)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    synth.write_fastq_pair(f1, f2, names, r1, r2)
    out = tmp_path / "out.fq"
    _run_cli(["pair", prefix, f1, f2, "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", "4", "-u", "750", "-F", "-nc"],
             out, batch_pairs=max(64, n_pairs // 5))
    got, want = fastq_records(str(out)), fastq_records(_oracle_stdout(prefix, f1, f2, L, tmp_path))
    assert len(got) == len(want) == 2 * n_pairs
    assert got == want


def test_md_text_of_alignments_trimmed_at_a_sequence_start(tmp_path):
    """Soak seed 4622, round 15 (2 x 250 bp, -P -b -p, two chunks, junk ends): four alignments across a sequence boundary whose LEFT part is dropped
    (BoundaryCheckDP) and began with soft-clipped bases.  The reference reads the MD text from `pos + trim` with trim counting read bases, the clipped ones
    included, i.e. up to a read length past the alignment's own span: the writer had only fetched span + 8 bases and printed 'N' beyond them."""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_parity.py"), "--rounds", "16", "--seed", "4622", "--only", "15"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    assert r.returncode == 0 and b"all 16 rounds equal" in r.stdout, r.stdout.decode()[-2000:]


@pytest.mark.parametrize("driver", ["cxx", "py"])
@pytest.mark.parametrize("tag,ini,flags", [("h", "soap4.ini", ["-F", "-nc"]), ("r", "soap4.ini", ["-F", "-P", "-nc", "-top", "100"]),
                                           ("n", "soap4-nt2.ini", ["-F", "-nc"])])
def test_cli_soap4_ini_stages_equal_reference(golden_dir, tmp_path, tag, ini, flags, driver):
    """The host-filter and 16S command lines of runMegaPath.sh:136,162 (`-C soap4.ini`: seeds of >= 22 bases, `-L 150`; 16S: `-F -P -nc
    -top 100`) against the real soap4's stdout (tests/golden/g5, make_golden_ini.py)."""
    d = os.path.join(golden_dir, "g5")
    out = tmp_path / "out.fq"
    _run_cli(["pair", os.path.join(golden_dir, "g1", "ref.fa.index"), os.path.join(d, "reads_1.fq"), os.path.join(d, "reads_2.fq"),
              "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", ini), "-L", "150", "-T", "2", "-u", "750"] + flags, out, driver=driver,
             batch_pairs=90 if driver == "cxx" else None)
    got, want = fastq_records(str(out)), fastq_records(os.path.join(d, "expected_%s.fq" % tag))
    assert len(got) == len(want) and got == want


def test_cli_two_worker_pools_on_one_device(golden_dir, tmp_path):
    """`-c 0,0`: the multi-device code path of the executable (one worker pool, one index replica and one set of contexts per listed
    device, a single ordered output) on real hardware -- both 'devices' are GPU 0 here.  Output = the real soap4's for golden a."""
    d = os.path.join(golden_dir, "g1")
    out = tmp_path / "out.fq"
    err = _run_cli(["pair", os.path.join(d, "ref.fa.index"), os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
                    "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "2", "-u", "750", "-F", "-nc", "-c", "0,0"], out, batch_pairs=17)
    assert fastq_records(str(out)) == fastq_records(os.path.join(d, "expected_a.fq"))
    assert b"2 GPU(s)" in err
