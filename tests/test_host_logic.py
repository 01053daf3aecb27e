"""CPU tests of the product's host side: C-ABI surface, read packing, host bookkeeping, and the N>1 path over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from megapath_amd import hipapi as H
from megapath_amd import pipeline, taxcount


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "megapath_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "megapath_amd", "csrc")])
    lib = C.CDLL(H.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(H.EXPORTS) == declared


def test_no_device_means_loud_failure_not_fallback():
    """Without a GPU the product must refuse to run: there is no CPU path behind the ABI."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = H.load_library()
    assert lib.mp_device_check(0, None, 0) == -19
    with pytest.raises(H.MpError):
        H.Index(os.path.join(ROOT, "tests", "golden", "g1", "ref.fa.index"), 0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "megapath_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle_lib" not in src and "mp_oracle" not in src and "libmp_oracle" not in src, fn


def test_pack_reads_matches_reference_packing_rules():
    # QueryParser.cpp:184-234 + IndexHandler.cpp:26-46: truncate to L-1, N->G, other letters->A, LSB-first 2 bits
    packed, lens, W = H.pack_reads([b"ACGTNRacgtu", b"T" * 40], 21)
    assert W == 2 and lens.tolist() == [11, 20]
    codes = [(int(packed[0, i // 16]) >> (2 * (i % 16))) & 3 for i in range(11)]
    assert codes == [0, 1, 2, 3, 2, 0, 0, 1, 2, 3, 3]
    assert all(((int(packed[1, i // 16]) >> (2 * (i % 16))) & 3) == 3 for i in range(20))


def test_cutoff_and_margin_rules():
    assert pipeline.dp_cutoff(150) == 30 and pipeline.dp_cutoff(300) == 60 and pipeline.margin(100) == 25 and pipeline.margin(101) == 30


def test_header_mapping_quirks():
    sc, v = pipeline._mapping_from_header(b"SCORE:300;300,seq1;290,seq2;", 0.95, 0.0)
    assert sc == 300 and v == [(300, b"300,seq1"), (290, b"290,seq2")]
    assert pipeline._mapping_from_header(b"IGNORE", 0.95, 0.0) == (0, [])
    assert pipeline._mapping_from_header(None, 0.95, 0.0) == (0, [])
    assert pipeline._mapping_from_header(b"SCORE:100;100,a;", 0.95, 200.0) == (100, [])


def test_shard_bounds_cover_all_pairs():
    for n, w in [(10, 3), (1 << 20, 8), (5, 8)]:
        cuts = [taxcount.shard_bounds(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from megapath_amd import taxcount
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g4 = os.path.join(sys.argv[1], "tests", "golden", "g4")
lines = open(os.path.join(g4, "expected.lsam.id"), "rb").read().splitlines(keepends=True)
pairs = len(lines) // 2
lo, hi = taxcount.shard_bounds(pairs, world, rank)            # contiguous block of pairs: 2 LSAM.id lines each
mine = lines[2 * lo:2 * hi]
total = taxcount.all_reduce_pass1(taxcount.pass1_from_lsam_id(mine, 40))
whole = taxcount.pass1_from_lsam_id(lines, 40)
assert total == whole, (total, whole)
expl = taxcount.explains(total)
want = sorted(l for l in open(os.path.join(g4, "expected.explains")).read().split("\n") if l)
assert sorted("%d explains %d" % e for e in expl) == want, (expl, want)
# pass 2 on this rank's shard; the shards in rank order are what the real reassign prints for the whole file
out = b"".join(taxcount.apply_explains(mine, expl))
ref = open(os.path.join(g4, "expected.ra.lsam.id"), "rb").read().splitlines(keepends=True)
assert out == b"".join(ref[2 * lo:2 * hi])
dist.destroy_process_group()
print("rank", rank, "ok", sorted(total.counts.items()))
'''


def test_two_rank_pass1_allreduce_equals_real_reassign(tmp_path):
    # N>1: the LSAM.id lines shard over 2 ranks by contiguous pair blocks; pass-1 counters (cc/reassign.cpp:64-110) are
    # all-reduced (dense vectors + gathered pair table); the derived `explains` relation equals the real reassign's and each
    # rank's pass-2 output equals its slice of the real reassign's stdout (tests/golden/g4, made by the reference tools).
    w = tmp_path / "worker.py"
    w.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", str(w), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_pass1_restatement_single_process(golden_dir):
    g4 = os.path.join(golden_dir, "g4")
    lines = open(os.path.join(g4, "expected.lsam.id"), "rb").read().splitlines(keepends=True)
    p = taxcount.pass1_from_lsam_id(lines, 40)
    assert p.counts and p.uniq and p.intersect                     # the fixture exercises all three counters
    assert sorted("%d explains %d" % e for e in taxcount.explains(p)) == sorted(l for l in open(os.path.join(g4, "expected.explains")).read().split("\n") if l)
    assert b"".join(taxcount.apply_explains(lines, taxcount.explains(p))) == open(os.path.join(g4, "expected.ra.lsam.id"), "rb").read()


def test_lsam_restatement_matches_reference_tool(golden_dir):
    """tests/lsam.py (the checker for the driver's -lsam mode) against the real cc/fastq2lsam's output on the goldens."""
    import os
    import lsam
    g = os.path.join(golden_dir, "g1")
    for tag, seq in (("a", True), ("d", True), ("e", True), ("g", True), ("h", True), ("a.noseq", False)):
        src = open(os.path.join(g, "expected_%s.fq" % tag.split(".")[0]), "rb").read()
        assert lsam.fastq_to_lsam(src, seq) == open(os.path.join(g, "expected_%s.lsam" % tag), "rb").read(), tag


def test_driver_plumbing_keeps_input_order_across_fake_gpus(tmp_path):
    """bin/soap4-mi355x with MP_PLUMBING_TEST (no device, nothing aligned): three 'GPUs' x two workers with jittered batch
    times and two 'chunks'; the printer must still write the batches -- contiguous blocks of pairs -- in input order."""
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "megapath_amd", "csrc")])
    d = os.path.join(ROOT, "tests", "golden", "g1")
    env = dict(os.environ, MP_PLUMBING_TEST="1", MP_PLUMBING_JITTER_US="4000", MP_BATCH_PAIRS="7")
    r = subprocess.run([exe, "pair", "x,y", os.path.join(d, "reads_a_1.fq"), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"),
                        "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", "151", "-T", "3", "-u", "750", "-F", "-nc", "-c", "0,1,2"],
                       env=env, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert b"NO ALIGNMENT IS DONE" in r.stderr and b"3 GPU(s) x 4 worker(s), 2 index chunk(s)" in r.stderr      # (four workers per GPU with several chunks, three with one)
    names = [l.split(b"\t")[0][1:] for l in r.stdout.split(b"\n") if l.startswith(b"@r")]
    src = [l.split()[0][1:-2] for l in open(os.path.join(d, "reads_a_1.fq"), "rb") if l.startswith(b"@r")]
    assert names[0::2] == src and names[1::2] == src


def test_driver_reports_corrupt_gzip_instead_of_truncating(tmp_path):
    import gzip
    d = os.path.join(ROOT, "tests", "golden", "g1")
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    raw = gzip.compress(open(os.path.join(d, "reads_a_1.fq"), "rb").read())
    bad = tmp_path / "bad_1.fq.gz"
    bad.write_bytes(raw[: len(raw) // 2])                       # truncated stream
    env = dict(os.environ, MP_PLUMBING_TEST="1")
    r = subprocess.run([exe, "pair", "x", str(bad), os.path.join(d, "reads_a_2.fq"), "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
                        "-L", "151", "-F", "-nc"], env=env, capture_output=True)
    assert r.returncode == 1 and b"error while reading" in r.stderr


def test_genome_cov_filter_restatement_matches_the_real_tool(tmp_path):
    """tests/spike_check.py::genome_cov_filter (the checker of the device spike filter) against cc/genomeCovFilter itself
    (oracle/_ref, built by oracle/Makefile.ref) on random coverage profiles."""
    import spike_check as S
    if not os.path.exists(S.REAL):
        pytest.skip("oracle/_ref/genomeCovFilter not built (needs /root/reference)")
    rng = np.random.default_rng(1)
    n_spikes = 0
    for trial in range(40):
        genome = [("s%d" % i, int(rng.integers(50, 3000))) for i in range(int(rng.integers(1, 5)))]
        bed = []
        for n, ln in genome:
            for _ in range(int(rng.integers(0, 60))):
                a = int(rng.integers(0, ln))
                b = min(ln, a + int(rng.integers(1, 200)))
                if rng.random() < 0.3:
                    a = int(ln * 0.4); b = min(ln, a + 100)
                bed.append((n, a, b, "r"))
        bg = S.bedgraph_bga(genome, bed)
        for k in (1, 2, 5):
            mine, _ = S.genome_cov_filter(genome, bg, k)
            assert S.real_genome_cov_filter(genome, bg, k, str(tmp_path)) == mine
            n_spikes += len(mine)
    assert n_spikes > 100


_RANK_CHILD = r'''
import json, os, sys
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert rank == int(os.environ["RANK"]) == int(os.environ["LOCAL_RANK"]) and world == int(os.environ["WORLD_SIZE"])
dist.barrier()
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)
print(json.dumps({"n_gpus": world, "rank": rank, "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
'''


def test_bench_gpus_n_launches_n_ranks_by_itself(tmp_path):
    """`python bench.py --gpus N` with no RANK in the environment must become N ranks: the launcher starts
    N children with the torchrun environment, relays rank 0's stdout as its own and fails when a rank fails."""
    child = tmp_path / "child.py"
    child.write_text(_RANK_CHILD)
    drv = tmp_path / "drv.py"
    drv.write_text("import sys; sys.path.insert(0, %r)\nimport bench\nraise SystemExit(bench.launch_ranks(int(sys.argv[1]), sys.argv[2:], script=%r))\n"
                   % (ROOT, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, str(drv), "3", "--steps", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line: rank 0's; the other ranks print to stderr
    import json
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 3, "rank": 0, "argv": ["--steps", "2"]}
    assert r.stderr.count('"n_gpus": 3') == 2
    r = subprocess.run([sys.executable, str(drv), "2", "--fail-rank", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 7, (r.returncode, r.stderr)
    # and bench.py itself takes that road before anything touches the GPU (here: no GPU at all, every rank says so and fails)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=600)
    # (the first rank to fail takes the other down: one or both say so)
    assert r.returncode != 0 and r.stderr.count("bench.py needs an MI355X") >= 1 and "rank" in r.stderr, r.stderr[-600:]


def _plumb(tmp_path, f1, f2, extra_env=None, L=151, threads=4, flags=("-F", "-nc")):
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    env = dict(os.environ, MP_PLUMBING_TEST="1", **(extra_env or {}))
    r = subprocess.run([exe, "pair", "x", str(f1), str(f2), "-o", str(tmp_path / "o"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"), "-L", str(L), "-T", str(threads),
                        "-u", "750", *flags], env=env, capture_output=True)
    return r


def _kseq_records(data: bytes):
    """kseq's reading of a FASTA/FASTQ text (klib kseq.h: sequence lines up to '+' or the next header, then quality lines until they
    are as long as the sequence) -> [(name, comment or None, seq, qual)]"""
    lines = data.replace(b"\r", b"").split(b"\n")
    i, out = 0, []
    while i < len(lines):
        if not lines[i]:
            i += 1
            continue
        assert lines[i][:1] in (b"@", b">")
        hdr = lines[i][1:]
        i += 1
        cut = len(hdr)
        for j, ch in enumerate(hdr):
            if ch in b" \t":
                cut = j
                break
        name, comment = hdr[:cut], (hdr[cut + 1:] if cut + 1 < len(hdr) else None)
        seq = b""
        while i < len(lines) and lines[i][:1] not in (b"+", b">", b"@"):
            seq += lines[i]
            i += 1
        qual = b""
        if i < len(lines) and lines[i][:1] == b"+":
            i += 1
            while i < len(lines) and len(qual) < len(seq):
                qual += lines[i]
                i += 1
        out.append((name, comment, seq, qual))
    return out


def _expected_plumbing(recs1, recs2, L):
    code = {c: "ACGT"[k] for k, c in enumerate("ACGT")}
    code.update({c.lower(): v for c, v in list(code.items())})
    code.update({"N": "G", "n": "G", "U": "T", "u": "T"})
    out = []
    for a, b in zip(recs1, recs2):
        for name, _, seq, qual in (a, b):
            nm = name.decode()
            if len(nm) > 2 and nm[-2] == "/" and nm[-1].isdigit():
                nm = nm[:-2]
            s = "".join(code.get(chr(c), "A") for c in seq[:L - 1])
            out.append("@%s\tSCORE:0;\n%s\n+\n%s\n" % (nm, s, qual[:len(s)].decode()))
    return "".join(out)


@pytest.mark.parametrize("shape", ["strict", "no_final_newline", "crlf", "blank_lines", "wrapped_fastq", "fasta", "wrapped_fasta", "lowercase_n", "gz", "gz_wrapped"])
@pytest.mark.parametrize("batch", [5, 1000])
def test_reader_shapes_equal_kseq_reading(tmp_path, shape, batch):
    """The ingest of the executable (mapped plain files cut into lines by several threads, gz inflated into the batch; strict fast
    path with a per-record check, careful kseq-style splitter otherwise) must hand over exactly the records kseq would
    (QueryParser.cpp:184-262): names without /1 /2, reads cut to L-1, N -> G, wrapped sequence / quality lines joined."""
    import gzip
    import random
    rng = random.Random(sum(map(ord, shape)))
    n = 37
    recs = [[], []]
    for k in range(2):
        for i in range(n):
            ln = rng.choice([30, 75, 150, 151, 160])
            alpha = "ACGT" if shape != "lowercase_n" else "ACGTNacgtnRY"
            seq = "".join(rng.choice(alpha) for _ in range(ln))
            qual = "".join(chr(rng.randrange(34, 74)) for _ in range(ln))
            if i % 5 == 0:
                qual = "@" + qual[1:]                     # a quality line that begins with '@'
            recs[k].append(("r%d/%d" % (i, k + 1), seq, qual))
    files = []
    for k in range(2):
        t = ""
        for i, (nm, seq, qual) in enumerate(recs[k]):
            fasta = shape in ("fasta", "wrapped_fasta")
            wrap = 60 if shape in ("wrapped_fastq", "wrapped_fasta", "gz_wrapped") else 10 ** 9
            body = "\n".join(seq[j:j + wrap] for j in range(0, len(seq), wrap))
            if fasta:
                t += ">%s\n%s\n" % (nm, body)
            else:
                qb = "\n".join(qual[j:j + wrap] for j in range(0, len(qual), wrap))
                t += "@%s\n%s\n+\n%s\n" % (nm, body, qb)
            if shape == "blank_lines" and i % 7 == 3:
                t += "\n"
        if shape == "no_final_newline":
            t = t[:-1]
        if shape == "crlf":
            t = t.replace("\n", "\r\n")
        data = t.encode()
        path = tmp_path / ("in_%d.fq" % (k + 1))
        if shape.startswith("gz"):
            path = tmp_path / ("in_%d.fq.gz" % (k + 1))
            path.write_bytes(gzip.compress(data))
        else:
            path.write_bytes(data)
        files.append((path, data))
    r = _plumb(tmp_path, files[0][0], files[1][0], {"MP_BATCH_PAIRS": str(batch)})
    assert r.returncode == 0, r.stderr.decode()
    want = _expected_plumbing(_kseq_records(files[0][1]), _kseq_records(files[1][1]), 151)
    assert r.stdout.decode() == want


def test_reader_rejects_garbage_and_mate_count_mismatch(tmp_path):
    a = tmp_path / "a.fq"; b = tmp_path / "b.fq"
    a.write_bytes(b"@r0/1\nACGT\n+\nIIII\n@r1/1\nACGT\n+\nIIII\n")
    b.write_bytes(b"@r0/2\nACGT\n+\nIIII\n")
    r = _plumb(tmp_path, a, b)
    assert r.returncode == 1 and b"not matched" in r.stderr
    b.write_bytes(b"@r0/2\nACGT\n+\nIIII\nACGT\nACGT\n")                 # a line that is neither header nor part of a record
    r = _plumb(tmp_path, a, b)
    assert r.returncode == 1 and b"does not begin with" in r.stderr
    b.write_bytes(b"@r0/2\nACGT\n+\nIIII\n@r1/2\nACGT\n+\n")               # cut short inside a record
    r = _plumb(tmp_path, a, b)
    assert r.returncode == 1 and b"ends inside" in r.stderr


@pytest.mark.parametrize("inflate", ["fast", "zlib", "parallel"])
def test_reader_gzip_members_levels_and_corruption(tmp_path, inflate):
    """The executable's own gzip decoder (host/fast_inflate.h; MP_INFLATE=zlib keeps zlib's gzread): concatenated members (bgzip, `cat a.gz b.gz`),
    an empty member, stored blocks (level 0), fixed-Huffman blocks, levels 1-9 -- the records must be what kseq reads from the plain text; a flipped
    bit or a truncated file must end the run with an error, never with different or fewer records."""
    import gzip
    import random
    import zlib
    rng = random.Random(77)
    texts = []
    for k in range(2):
        t = "".join("@r%d/%d\n%s\n+\n%s\n" % (i, k + 1, "".join(rng.choice("ACGT") for _ in range(150)), "".join(chr(rng.randrange(35, 74)) for _ in range(150))) for i in range(3000))
        texts.append(t.encode())
    env = {"MP_BATCH_PAIRS": "700"}
    if inflate == "zlib":
        env["MP_INFLATE"] = "zlib"
    if inflate == "parallel":                                             # several threads on the one stream (host/par_gzip.h), segments of 64 KiB
        env.update(MP_PARGZ_MIN="0", MP_PARGZ_SEG="65536", MP_INFLATE_THREADS="4")
    else:
        env["MP_PARGZ_MIN"] = str(1 << 40)
    want = _expected_plumbing(_kseq_records(texts[0]), _kseq_records(texts[1]), 151)

    def members(data, k):
        cuts = sorted(rng.sample(range(1, len(data)), 5))
        parts = [data[a:b] for a, b in zip([0] + cuts, cuts + [len(data)])]
        out = b""
        for i, part in enumerate(parts):
            if i == 2:
                out += gzip.compress(b"")                                     # an empty member in between
            if i == 3:
                c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)   # fixed Huffman codes
                out += c.compress(part) + c.flush()
            else:
                out += gzip.compress(part, [0, 1, 6, 9, 3, 0][(i + k) % 6])
        return out
    files = []
    for k in range(2):
        f = tmp_path / ("m_%d.fq.gz" % (k + 1))
        f.write_bytes(members(texts[k], k))
        files.append(f)
    r = _plumb(tmp_path, files[0], files[1], env)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == want
    # one flipped bit somewhere in the compressed data of file 1: error (CRC at the latest)
    raw = bytearray(files[0].read_bytes())
    for trial in range(6):
        bad = bytearray(raw)
        i = rng.randrange(40, len(bad) - 40)
        bad[i] ^= 1 << rng.randrange(8)
        b1 = tmp_path / "bad_1.fq.gz"
        b1.write_bytes(bytes(bad))
        r = _plumb(tmp_path, b1, files[1], env)
        assert r.returncode == 1 and b"error while reading" in r.stderr, (trial, i, r.returncode, r.stderr[-300:])
    # truncated in the middle and inside the trailer
    for cut in (len(raw) // 3, len(raw) - 5):
        b1 = tmp_path / "cut_1.fq.gz"
        b1.write_bytes(bytes(raw[:cut]))
        r = _plumb(tmp_path, b1, files[1], env)
        assert r.returncode == 1 and (b"error while reading" in r.stderr or b"ends inside" in r.stderr or b"not matched" in r.stderr), r.stderr[-300:]


def _bgzf(data: bytes, block: int = 60000) -> bytes:
    """htslib's blocked gzip: every member has the 'BC' extra field with its own size; an empty member marks the end."""
    import struct
    import zlib
    out = b""
    for i in list(range(0, len(data), block)) + [None]:
        raw = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(raw), len(raw))
    return out


@pytest.mark.parametrize("threads", ["3", "0"])
def test_reader_bgzf_members_inflate_side_by_side(tmp_path, threads):
    """BGZF input (bgzip; what BBTools writes when bgzip is installed): the members are inflated by several threads and handed over in file
    order (MP_INFLATE_THREADS=0: one after the other, like any gzip file).  Records as kseq reads the plain text; a flipped bit is an error."""
    import random
    rng = random.Random(5)
    texts = []
    for k in range(2):
        t = "".join("@q%d/%d\n%s\n+\n%s\n" % (i, k + 1, "".join(rng.choice("ACGT") for _ in range(150)), "".join(rng.choice("F:,#") for _ in range(150))) for i in range(20000))
        texts.append(t.encode())
    files = []
    for k in range(2):
        f = tmp_path / ("b_%d.fq.gz" % (k + 1))
        f.write_bytes(_bgzf(texts[k], 60000 if k == 0 else 9000))
        files.append(f)
    env = {"MP_BATCH_PAIRS": "3500", "MP_INFLATE_THREADS": threads}
    r = _plumb(tmp_path, files[0], files[1], env)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == _expected_plumbing(_kseq_records(texts[0]), _kseq_records(texts[1]), 151)
    raw = bytearray(files[0].read_bytes())
    for i in (len(raw) // 2, len(raw) // 3 + 7, 3 * len(raw) // 4):
        bad = bytearray(raw)
        if bad[i - 18:i - 14] == b"\x1f\x8b\x08\x04":
            i += 40                                                       # (not a member's time stamp / OS byte: those are not checked by anybody)
        bad[i] ^= 0x10
        b1 = tmp_path / "badb_1.fq.gz"
        b1.write_bytes(bytes(bad))
        r = _plumb(tmp_path, b1, files[1], env)
        assert r.returncode == 1 and b"error while reading" in r.stderr, (i, r.returncode, r.stderr[-300:])


def _pargz_tool(tmp_path):
    exe = tmp_path / "pargz_check"
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "tools", "pargz_check.cpp"), "-lz"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    return str(exe)


def test_parallel_gzip_reader_equals_zlib_on_every_stream_shape(tmp_path):
    """host/par_gzip.h (several threads on ONE deflate stream: block starts found by search, segments decoded without their window into
    16-bit symbols, markers resolved in file order) against zlib's gzread, byte for byte: compression levels, concatenated members, fixed
    and stored blocks, full-flush points, long matches (a window full of markers), tiny members, input that is not text (no start is ever
    found: one thread decodes everything).  Also the folding CRC-32 (host/crc32_fold.h) is what checks every member here."""
    import gzip
    import random
    import zlib
    exe = _pargz_tool(tmp_path)
    rng = random.Random(9)

    def fastq(n, tag):
        return "".join("@%s%d\n%s\n+\n%s\n" % (tag, i, "".join(rng.choice("ACGT") for _ in range(150)), "".join(rng.choice("FFFFF:,#") for _ in range(150))) for i in range(n)).encode()
    base = fastq(12000, "r")
    shapes = {}
    for lv in (1, 6, 9):
        shapes["level%d" % lv] = gzip.compress(base, lv)
    shapes["members"] = b"".join(gzip.compress(base[a:a + 700000], [6, 1, 9, 0][k % 4]) for k, a in enumerate(range(0, len(base), 700000))) + gzip.compress(b"")
    c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    shapes["fixed"] = c.compress(base) + c.flush()
    shapes["stored"] = gzip.compress(base, 0)
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    shapes["flushes"] = b"".join(c.compress(base[a:a + 200000]) + c.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH) for k, a in enumerate(range(0, len(base), 200000))) + c.flush()
    rep = fastq(40, "x")
    shapes["repeats"] = gzip.compress(rep * 3000 + base[:500000] + rep * 3000, 6)          # matches that reach back 32 KiB again and again: markers throughout a segment
    shapes["binary"] = gzip.compress(bytes(rng.randrange(256) for _ in range(300000)) + base[:2000000], 6)
    shapes["tiny"] = gzip.compress(b"@a\nA\n+\nF\n")
    one = fastq(1, "same")
    shapes["ratio"] = gzip.compress(one * 150000, 6)                      # inflates 150-fold: a segment hands over what it has before it is done (bounded memory)
    shapes["stored_big"] = gzip.compress(base * 6, 0)                     # no dynamic block anywhere: the search gives up, one thread decodes, in parts
    for name, data in shapes.items():
        f = tmp_path / (name + ".gz")
        f.write_bytes(data)
        for threads, seg in ((4, 65536), (3, 262144)):
            r = subprocess.run([exe, str(f), str(threads), str(seg)], capture_output=True)
            assert r.returncode == 0 and b"DIFFERENT" not in r.stdout, (name, threads, seg, r.stdout.decode()[-600:], r.stderr.decode()[-300:])
    # the search does find starts in ordinary files (otherwise everything above ran on one thread)
    r = subprocess.run([exe, str(tmp_path / "level6.gz"), "4", "65536"], capture_output=True)
    import re
    m = re.search(rb"(\d+) segments, (\d+) started without their window", r.stdout)
    assert m and int(m.group(2)) >= int(m.group(1)) // 2, r.stdout


def test_parallel_gzip_reader_survives_false_starts(tmp_path):
    """A start that is none (the chain's decoder runs past it without standing on it) voids its segment: the segment's decoder may be done
    long before the chain finds that out, and its slot and buffer must still come back -- otherwise the reader stops handing out pieces
    after `ahead` such segments (advisor finding of round 3).  MP_PARGZ_TEST_BOGUS_START=n plants a bogus start bit in every n-th segment;
    output equal to zlib's, no slot left claimed, however many such segments go by -- and a decoder that passes three of them in a row is
    no longer reported as a corrupt stream."""
    import gzip
    import random
    import re
    exe = _pargz_tool(tmp_path)
    rng = random.Random(11)
    data = "".join("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(150)), "".join(rng.choice("FFFFF:,#") for _ in range(150))) for i in range(30000)).encode()
    f = tmp_path / "fs.gz"
    f.write_bytes(gzip.compress(data, 6))
    n_false = {}
    for every in (2, 3, 1):                        # 1: every segment but the first has a bogus start -- one decoder passes them all
        for threads, seg in ((4, 65536), (2, 131072)):
            r = subprocess.run([exe, str(f), str(threads), str(seg)], capture_output=True, env=dict(os.environ, MP_PARGZ_TEST_BOGUS_START=str(every)), timeout=120)
            out = r.stdout.decode()
            assert r.returncode == 0 and "DIFFERENT" not in out, (every, threads, seg, out[-600:], r.stderr.decode()[-300:])
            assert "segments still holding a slot after the last piece: 0" in out, out[-400:]
            m = re.search(r"(\d+) segments, (\d+) started without their window, (\d+) false starts", out)
            n_false[(every, threads)] = (int(m.group(1)), int(m.group(3)))
    segs, false_starts = n_false[(2, 4)]
    assert false_starts >= segs // 2 - 1 > 6, n_false       # many more than the reader's look-ahead (threads + 2)


@pytest.mark.parametrize("shape", ["gzip_members", "bgzf"])
def test_reader_rejects_a_match_that_reaches_into_the_member_before(tmp_path, shape):
    """A deflate match must not reach back past the start of ITS gzip member, even though the member before it left its output in the same
    buffer (concatenated members, the groups of a BGZF file): zlib says "invalid distance too far back"; so does the reader's own decoder,
    before the CRC has a say (advisor finding of round 3).  The second member is deflated with the first one's text as its preset
    dictionary, i.e. its matches refer to bytes of the member before; CRC-32 and length in its trailer are those of the text a decoder
    that accepted the references would produce -- only the distance check can refuse it."""
    import random
    import struct
    import zlib
    rng = random.Random(17)

    def fastq(n, tag, k):
        return "".join("@%s%d/%d\n%s\n+\n%s\n" % (tag, i, k, "".join(rng.choice("ACGT") for _ in range(150)), "F" * 150) for i in range(n)).encode()
    first = fastq(150, "a", 1)
    second = first[first.index(b"@a70/"):]          # the same records again: every byte is a match into the member before
    assert len(second) < 32768
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=first[-32768:])
    body = c.compress(second) + c.flush()
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(body)    # zlib itself: invalid distance too far back
    def member(raw_deflate, text):
        if shape == "bgzf":
            return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw_deflate) + 25) + raw_deflate + struct.pack("<II", zlib.crc32(text), len(text))
        return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + raw_deflate + struct.pack("<II", zlib.crc32(text), len(text))
    c0 = zlib.compressobj(6, zlib.DEFLATED, -15)
    good = member(c0.compress(first) + c0.flush(), first)
    empty = member(zlib.compressobj(6, zlib.DEFLATED, -15).flush(), b"") if shape == "bgzf" else b""
    f1 = tmp_path / "x_1.fq.gz"
    f1.write_bytes(good + member(body, second) + empty)
    mate = (first + second).replace(b"/1\n", b"/2\n")
    f2 = tmp_path / "x_2.fq"
    f2.write_bytes(mate)
    for threads in ("3", "0"):
        r = _plumb(tmp_path, f1, f2, {"MP_INFLATE_THREADS": threads})
        assert r.returncode == 1 and b"error while reading" in r.stderr and b"too far back" in r.stderr, (threads, r.returncode, r.stderr[-300:])
        assert r.stdout == b"", "records of a corrupt file were printed"
