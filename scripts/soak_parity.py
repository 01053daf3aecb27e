"""Randomised whole-path parity campaign on the GPU box: bin/soap4-mi355x against the oracle (test infrastructure) on freshly generated
genomes and read sets with randomly drawn shapes.  python scripts/soak_parity.py [--rounds 40] [--seed 1]; prints every configuration and
stops with exit code 1 at the first difference (the configuration is then reproducible from its line)."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from conftest import fastq_records  # noqa: E402
from megapath_amd import index_builder, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--inis", default="both", help="both (drawn per round), soap4.ini or soap4-nt2.ini")
ap.add_argument("--only", type=int, default=None, help="run only this round of the seed's sequence (the rounds before it are drawn and skipped)")
ap.add_argument("--env", default="", help="K=V,K=V: switches that override the round's drawn ones (to bisect a differing round)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
lib = O.oracle()


def set_ini(p, ini):
    if ini == "soap4.ini":             # the two keys in which the shipped ini files differ (mmpSeedMinLength, mmpReseedLen)
        p.mmp.seed_min_length, p.mmp.reseed_len = 22, 23


for rnd in range(args.rounds):
    read_len = int(rng.choice([76, 100, 150, 150, 150, 250, 300]))
    cfg = dict(n_bases=int(rng.integers(100000, 800000)), n_seqs=int(rng.integers(1, 8)), rep=float(rng.choice([0.0, 0.05, 0.2, 0.5])),
               n_pairs=int(rng.integers(500, 4000)), read_len=read_len, gseed=int(rng.integers(1, 1 << 30)), rseed=int(rng.integers(1, 1 << 30)),
               mode=int(rng.choice([1, 1, 2])), threads=int(rng.integers(1, 6)), bam=bool(rng.random() < 0.4 and read_len <= 300),
               # the database cut into 1-4 index chunks resident at once (the NT loop in one process): the oracle is chained chunk after chunk,
               # each run reading the previous run's stdout (runMegaPath.sh:191-227); 0 = one index, no chain
               chunks=int(rng.choice([0, 0, 2, 3, 4])), host_merge=bool(rng.random() < 0.3),
               # soap4.ini (seeds of >= 22 bases: the pre-filter then rules a start out by any of six 17-mers) or soap4-nt2.ini (17)
               ini=str(rng.choice(["soap4-nt2.ini", "soap4-nt2.ini", "soap4.ini"])) if args.inis == "both" else args.inis,
               # which optional index tables the run may build (every combination takes another path through the seeding kernels)
               env={k: str(int(v)) for k, v in (("MP_KMER_FILTER", rng.choice([0, 1, 1, 1, 16])), ("MP_KMER15", rng.random() < 0.7), ("MP_LKT16", rng.random() < 0.7),
                                                ("MP_SA_DENSE", rng.random() < 0.7), ("MP_LKT16U", rng.random() < 0.7), ("MP_SA_DENSE_SPACING", rng.choice([1, 1, 2, 4])),
                                                ("MP_DP_DIAG", rng.random() < 0.8), ("MP_DP_ONE_MISMATCH", rng.choice([0, 1, 2, 2, 2])), ("MP_DP_RESCUE_BAND", rng.random() < 0.8), ("MP_MERGE_ROWS", rng.random() < 0.8),
                                                # round 4: the certificate's variants for short reads / mate-rescue windows; -b with resident chunks through the device reduction or every chunk's lists
                                                ("MP_DP_THERMO_MORE", rng.random() < 0.75), ("MP_BAM_HOST_MERGE", rng.random() < 0.3),
                                                # the -top rule across resident chunks (two passes, pairs that cannot reach top% of their best known score lose their rescue work) on / off
                                                ("MP_TOP_FLOOR", rng.random() < 0.7), ("MP_TOP_FLOOR_MATE", rng.random() < 0.75),
                                                # round 4, late: the 13-mer table as 8-byte entries with a seed's first extension step in them; the stash between the two passes as 16-byte records
                                                ("MP_LKT14", rng.random() < 0.5), ("MP_STASH_PLAIN", rng.random() < 0.25))},
               # chunked databases: besides the copied repeats, every chunk may get a DIVERGED copy of a stretch of another chunk's sequence -- entries near and
               # below top% of a read's best score in other chunks, which is what the -top rule across chunks decides about
               div_copy=float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.2])), div_seed=int(rng.integers(1, 1 << 30)),
               kw=dict(sub=float(rng.choice([0.0, 0.005, 0.01, 0.03, 0.06])), ins=float(rng.choice([0.0, 0.001, 0.004])), dele=float(rng.choice([0.0, 0.001, 0.004])),
                       random_frac=float(rng.choice([0.0, 0.05, 0.3])), broken_mate_frac=float(rng.choice([0.0, 0.1, 0.3])),
                       noisy_mate_frac=float(rng.choice([0.0, 0.2])), noisy_sub=0.15, n_frac=float(rng.choice([0.0, 0.002])),
                       long_indel_frac=float(rng.choice([0.0, 0.2])), long_indel_max=int(rng.integers(4, 27)),
                       junk_end_frac=float(rng.choice([0.0, 0.3])), junk_end_max=int(rng.integers(5, 60)),
                       len_range=(max(20, read_len // 3), read_len) if rng.random() < 0.4 else None,
                       frag_mean=read_len + 200, frag_sd=40, frag_min=read_len + 20, frag_max=min(740, read_len + 420)))
    if args.only is not None and rnd != args.only:
        continue
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("=", 1)
        cfg["env"][k] = v
    print("round %d: %r" % (rnd, cfg), flush=True)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        n_chunks = cfg["chunks"]
        if cfg["env"].get("MP_BAM_HOST_MERGE") == "0":
            del cfg["env"]["MP_BAM_HOST_MERGE"]           # (the switch is "set or not set")
        genome = synth.random_genome(cfg["n_bases"], cfg["gseed"], n_seqs=max(cfg["n_seqs"], n_chunks), repeat_frac=cfg["rep"])
        prefix = os.path.join(tmp, "g.index")
        if n_chunks:
            # sequences dealt round-robin to the chunks (copied repeats then cross chunk borders: one read, hits in several chunks)
            parts = [[genome[i] for i in range(len(genome)) if i % n_chunks == c] for c in range(n_chunks)]
            if cfg["div_copy"] > 0:
                drng = np.random.default_rng(cfg["div_seed"])
                letters = np.frombuffer(b"ACGT", dtype=np.uint8)
                for c in range(n_chunks):
                    src = parts[(c + 1) % n_chunks][0][1]
                    n = int(min(len(src), drng.integers(20000, 120000)))
                    a = int(drng.integers(0, len(src) - n + 1))
                    seg = np.array(src[a:a + n], dtype=np.uint8, copy=True)
                    k = (drng.random(n) < cfg["div_copy"]) & np.isin(seg, letters)
                    seg[k] = letters[(np.searchsorted(letters, seg[k]) + drng.integers(1, 4, size=int(k.sum()))) & 3]
                    seg = seg[drng.random(n) >= 0.002]                                  # a few deletions
                    parts[c] = parts[c] + [("divcopy%d" % c, seg)]
            prefixes = []
            for c, part in enumerate(parts):
                pc = os.path.join(tmp, "c%d.index" % c)
                index_builder.write_index_files(pc, index_builder.build_index_arrays(part, sa_interval=16))
                prefixes.append(pc)
            prefix = ",".join(prefixes)
        else:
            index_builder.write_index_files(prefix, index_builder.build_index_arrays(genome, sa_interval=16))
        names, r1, r2 = synth.simulate_pairs(genome, cfg["n_pairs"], read_len, cfg["rseed"], **cfg["kw"])
        f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
        synth.write_fastq_pair(f1, f2, names, r1, r2)
        L = read_len + 1
        out = os.path.join(tmp, "out.fq")
        cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefix, f1, f2, "-o", os.path.join(tmp, "o"), "-C", os.path.join(ROOT, "tests", cfg["ini"]),
               "-L", str(L), "-T", str(cfg["threads"]), "-u", "750", "-F" if cfg["mode"] == 1 else "-P", "-nc"] + (["-b", "-p", "-D", "rg1", "-A", "default"] if cfg["bam"] else [])
        env = dict(os.environ, MP_BATCH_PAIRS=str(max(64, cfg["n_pairs"] // 3)), **cfg["env"])
        if cfg["host_merge"]:
            env["MP_HOST_MERGE"] = "1"
        with open(out, "wb") as fo:
            r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
        if r.returncode != 0:
            print("FAILED to run:", r.stderr.decode()[-800:])
            sys.exit(1)
        want = os.path.join(tmp, "want.fq")
        if n_chunks:
            in1, in2 = f1, f2
            for c, pc in enumerate(prefixes):
                ix = lib.mpo_index_open(pc.encode())
                p = O.default_params(L=L, mode=cfg["mode"], nc=(c == 0))
                set_ini(p, cfg["ini"])
                st = O.AlignStats()
                if cfg["bam"]:      # every run of the chain writes the same -o prefix (runMegaPath.sh:199): the last chunk's BAM records remain
                    assert lib.mpo_align_fastq_bam(ix, C.byref(p), in1.encode(), in2.encode(), want.encode(), None, os.path.join(tmp, "obam").encode(), 1, b"rg1", C.byref(st)) == 0
                else:
                    assert lib.mpo_align_fastq(ix, C.byref(p), in1.encode(), in2.encode(), want.encode(), None, C.byref(st)) == 0
                lib.mpo_index_close(ix)
                if c + 1 < n_chunks:
                    lines = open(want, "rb").read().split(b"\n")
                    in1, in2 = os.path.join(tmp, "chain_%d_1.fq" % c), os.path.join(tmp, "chain_%d_2.fq" % c)
                    with open(in1, "wb") as fa, open(in2, "wb") as fb:
                        for i in range(0, len(lines) - 7, 8):
                            fa.write(b"\n".join(lines[i:i + 4]) + b"\n")
                            fb.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")
            ix = None
        else:
            ix = lib.mpo_index_open(prefix.encode())
            p = O.default_params(L=L, mode=cfg["mode"], nc=True)
            set_ini(p, cfg["ini"])
            st = O.AlignStats()
        if n_chunks:
            pass
        elif cfg["bam"]:
            assert lib.mpo_align_fastq_bam(ix, C.byref(p), f1.encode(), f2.encode(), want.encode(), None, os.path.join(tmp, "obam").encode(), 1, b"rg1", C.byref(st)) == 0
        else:
            assert lib.mpo_align_fastq(ix, C.byref(p), f1.encode(), f2.encode(), want.encode(), None, C.byref(st)) == 0
        if ix is not None:
            lib.mpo_index_close(ix)
        if cfg["bam"]:
            import bamtext
            for kind, path in (("dpout", "o.dpout.1"), ("gout", "o.gout.1"), ("unpair", "o.unpair")):
                got_recs = sorted(bamtext.read_bam(os.path.join(tmp, path))[2])
                want_recs = sorted(open(os.path.join(tmp, "obam.%s.txt" % kind)).read().splitlines())
                if got_recs != want_recs:
                    bad = [i for i in range(min(len(got_recs), len(want_recs))) if got_recs[i] != want_recs[i]]
                    print("BAM %s DIFFERENT: %d vs %d records" % (kind, len(got_recs), len(want_recs)))
                    print(" %d records differ" % len(bad))
                    for i in bad[:4]:
                        gf, wf = got_recs[i].split("\t"), want_recs[i].split("\t")
                        print(" record", gf[:9], "fields that differ:", [(k, a, b) for k, (a, b) in enumerate(zip(gf, wf)) if a != b][:6], len(gf), len(wf))
                    sys.exit(1)
        a, b = fastq_records(out), fastq_records(want)
        if a != b:
            bad = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
            print("DIFFERENT: %d vs %d records, first differing %s" % (len(a), len(b), bad[:5]))
            for i in bad[:3]:
                print(" got ", a[i][:3]); print(" want", b[i][:3])
            sys.exit(1)
print("all %d rounds equal" % args.rounds)
