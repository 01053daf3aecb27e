#!/usr/bin/env python3
"""bench.py -- paired 150 bp reads aligned per second against a 2BWT index on MI355X (BASELINE.json metric).

One process per GPU (`torchrun`-style env: RANK / LOCAL_RANK / WORLD_SIZE).  Reads shard across ranks, the index is
replicated into every GPU's HBM, no data-path collective: `"scaling": "weak"`; the per-reference
hit counters of the step are what the ranks all-reduce afterwards (RCCL), as the `reassign` pass 1 they feed would.

A *step* is one pass of the whole hot path over one batch of synthetic pairs: upload of the packed reads (pinned host
memory -> HBM), deep-DP round (MMP seeding, locate, merge, pairing, both DP legs with traceback, hit assembly), rescue
round (single-end DP, mate rescue) and the arrival of all three hit lists in pinned host memory.

Prints ONE JSON line on rank 0.  Besides the contract fields:
  roofline        seed stage (K1 + K2), priced with the bytes of the events that really load memory; `traffic` = HBM bytes
                  per launch from rocprofv3 counters collected LIVE in two child runs of this script (FETCH_SIZE,
                  WRITE_SIZE in separate passes); the work the short cuts stand for is in `equivalent_work`, not in `frac`
  dp              cells the DP kernels really compute (device census of one untimed step) over their time
  cpu_baseline    the real reference soap4 (oracle/_ref, when it travelled with the snapshot) on all host cores
  parity_sample   the same 100 k pairs through the product executable, record by record against that soap4's stdout
  legs            C3 / C4 / C5 workload shapes, two steps each, with their kernel split
"""
from __future__ import annotations

import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED_STAGE_KERNELS = ("seed_kernel", "seed_overflow_kernel", "expand_hits_kernel", "locate_kernel")
PMC_REGEX = "gather64_kernel|seed_kernel|seed_overflow_kernel|locate_kernel|expand_hits_kernel"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


NT2_INI = """[Database]
SaValueFileExt=.sa
[Alignment]
MaxOutputPerRead=200
Soap3MisMatchAllow=1
[PairEnd]
StrandArrangement=+/-
MaxOutputPerPair=200
MaxHitsEachEndForPairing=8000
[DP]
MatchScore=1
MismatchScore=-2
GapOpenScore=-3
GapExtendScore=-1
DPScoreThreshold=30
NumberOfRoundOfDeepDPForShortReads=1
NumberOfRoundOfDeepDPForLongReads=1
[Score]
MinMAPQ=1
MaxMAPQ=60
BWALikeScore=1
[Index]
ShareIndex=0
[Reads]
MaxLenReadName=128
[Clipping]
MaxFrontLenClipped=130
MaxEndLenClipped=130
[OtherSettings]
SkipSOAP3Alignment=1
SkipSingleEndDP=1
SkipDefaultDP=0
ProceedDPForTooManyHits=0
[SeedingRound1ForShortReads]
mmpSeedingScheme=1
[SeedingRound1ForLongReads]
mmpSeedingScheme=1
[SingleEndDP]
DPScoreThreshold=30
[DefaultDP]
DPScoreThreshold=30
[MMP]
mmpSeedSAsizeThreshold=30
mmpSeedMinLength=17
mmpUniqThreshold=6
mmpIndelFuzz=5
mmpGoodSeedLen=27
mmpReseedLen=18
mmpReseedRLTratio=0.7
mmpReseedAbsDiff=4
mmpShortSeedRatio=0.5
"""


HG_INI = NT2_INI.replace("mmpSeedMinLength=17", "mmpSeedMinLength=22").replace("mmpReseedLen=18", "mmpReseedLen=23")   # soap4.ini
INIS = {"soap4.ini": HG_INI, "soap4-nt2.ini": NT2_INI}


def mmp_of(ini_name):
    from megapath_amd import hipapi as H
    return H.MmpParams.human() if ini_name == "soap4.ini" else H.MmpParams.nt2()


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """A batch of synthetic pairs in pinned host memory (what a reader thread hands to the device stage)."""

    def __init__(self, text, n_pairs, seed, read_len=150, L=None, **sim):
        from megapath_amd import simulate
        codes, lens = simulate.paired_reads(text, n_pairs, read_len, seed=seed, **sim)
        self.L = L if L is not None else int(codes.shape[1]) + 1      # the -L value: maxReadLength, reads are cut to L - 1 (QueryParser.cpp:188)
        self.W = (self.L + 15) // 16
        self.codes = codes                                            # untruncated: what the FASTQ files of the parity sample hold
        if codes.shape[1] > self.L - 1:
            codes = codes[:, :self.L - 1].contiguous()
            lens = lens.clamp(max=self.L - 1)
        self.max_len = int(codes.shape[1])                            # longest read as the aligner sees it
        packed = simulate.pack_codes(codes, self.W)
        self.n_pairs = n_pairs
        self.sum_len = int(lens.sum().item())
        self.packed_h = packed.cpu().pin_memory()
        self.lens_h = lens.cpu().pin_memory()
        self.packed_np = self.packed_h.numpy().view(np.uint32)
        self.lens_np = self.lens_h.numpy().view(np.uint32)
        del packed


class WorkloadSlice:
    """A contiguous block of a workload's pairs (same pinned arrays, no copy): what one of several contexts works on."""

    def __init__(self, wl, lo_pair, hi_pair):
        self.packed_np = wl.packed_np[2 * lo_pair:2 * hi_pair]
        self.lens_np = wl.lens_np[2 * lo_pair:2 * hi_pair]
        self.n_pairs = hi_pair - lo_pair
        self.max_len, self.L, self.W = wl.max_len, wl.L, wl.W


def run_overlapped(ctxs, slices, params, n_steps, barrier=None):
    """The same batch cut into len(ctxs) contiguous blocks, one context (own HIP streams) and one host thread per block: the
    latency-bound seeding of one block overlaps with the VALU-bound DP of another.  Returns the wall time of n_steps steps."""
    import threading
    errs = []

    def worker(i):
        try:
            for _ in range(n_steps):
                step(ctxs[i], slices[i], params)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(ctxs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    if errs:
        raise errs[0]
    return time.perf_counter() - t0


def step(ctx, wl, params, fetch_all=True):
    """One pass of the hot path over a batch (see the module docstring)."""
    ctx.set_reads(wl.packed_np, wl.lens_np)                      # H2D of the packed reads
    _, nh, st = ctx.deep_dp_round(params, fetch=False)           # the pair hits start travelling to the host ...
    _, _, rst = ctx.rescue_round(params, False, fetch=False)     # ... while the rescue passes run; their hits follow
    hits, nh = ctx.pair_hits_view()                              # the step ends when all three lists have arrived
    singles = resc = None
    if fetch_all:
        singles, ns = ctx.single_hits_view()
        resc, nr = ctx.rescue_hits_view()
    return hits, nh, st, rst


def run_steps(ctx, wl, params, n_steps, barrier=None):
    acc = {}

    def add(k, v):
        acc[k] = acc.get(k, 0) + v
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        _, nh, st, rst = step(ctx, wl, params)
        for k, v in (("seed", st.seed.ms_seed), ("locate", st.seed.ms_locate), ("merge", st.seed.ms_merge), ("pair", st.ms_pair),
                     ("fill", st.ms_dp_fill), ("trace", st.ms_dp_trace), ("rescue_tasks", rst.ms_tasks), ("rescue_fill", rst.ms_dp_fill),
                     ("rescue_trace", rst.ms_dp_trace)):
            add("ms_" + k, v)
        s = st.seed
        for k in ("n_occ", "n_occ_loaded", "n_lkt", "n_lkt16", "n_sa", "n_seedpos", "n_prefilter", "n_prefilter_loads", "n_text_words", "n_steps_skipped",
                  "n_seeds", "n_hits", "n_sa_seed_kernel"):
            add(k, int(getattr(s, k)))
        add("cand", int(st.n_candidates)); add("pair_hits", int(st.n_pair_hits))
        add("failed", int(rst.n_failed_pairs)); add("stasks", int(rst.n_single_tasks)); add("rtasks", int(rst.n_rescue_tasks))
        add("single_hits", int(rst.n_single_hits)); add("rescue_hits", int(rst.n_rescue_hits))
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return time.perf_counter() - t0, acc


def loaded_bytes(acc, sum_len_per_step, n_steps, fused=False):
    """Bytes the seed stage's kernels request from memory, by event (each counted by the kernels themselves):
    64 B per rank block loaded, 8 B per look-up in the 13-mer / 16-mer tables (two adjacent 32-bit bounds), 8 B per pre-filter
    entry loaded, 4 B per suffix-array entry, 8 B per 16-base text word compared, 0.25 B per read base (2-bit reads),
    16 B per kept seed position written.  `fused` (whole suffix array resident: no expand / locate launch, the merge kernel reads the
    one suffix-array entry per hit that is still needed): only the seeding kernel's own suffix-array reads are counted, because only
    its time is in the denominator."""
    n_sa = acc["n_sa_seed_kernel"] if fused else acc["n_sa"]
    return (64 * acc["n_occ_loaded"] + 8 * (acc["n_lkt"] + acc["n_lkt16"]) + 8 * acc["n_prefilter_loads"] + 4 * n_sa + 8 * acc["n_text_words"]
            + 0.25 * sum_len_per_step * n_steps + 16 * acc["n_seedpos"])


# ------------------------------------------------------------------------------------------------ PMC child runs
def pmc_pass(counter, args, tmpdir):
    """Runs this script once more under rocprofv3 with ONE counter (separate passes, MI355X_MICROARCH.md) and returns
    {kernel: [value per launch]}.  The child does one step of the same workload; its last launch of each seed-stage kernel is
    the timed configuration."""
    out_dir = os.path.join(tmpdir, "pmc_" + counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-include-regex", PMC_REGEX, "--output-format", "csv", "-d", out_dir, "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "0", "--genome-mbp", str(args.genome_mbp),
           "--pairs-per-step", str(args.pairs_per_step), "--read-len", str(args.read_len), "--ini", args.ini] + (["--L", str(args.L)] if args.L is not None else [])
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if r.returncode != 0:
        log("[bench] rocprofv3 --pmc %s child failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-400:]))
        return None
    per = {}
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter:
                name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()       # "void seed_kernel<1>(...)" -> "seed_kernel"
                per.setdefault(name, []).append(float(row["Counter_Value"]))
    log("[bench] PMC pass %s: %.0f s, kernels %s" % (counter, time.time() - t0, {k: len(v) for k, v in per.items()}))
    return per


def live_traffic(args, tmpdir, kernels=SEED_STAGE_KERNELS):
    if shutil.which("rocprofv3") is None:
        return None
    fetch = pmc_pass("FETCH_SIZE", args, tmpdir)
    write = pmc_pass("WRITE_SIZE", args, tmpdir) if fetch else None
    if not fetch or not write:
        return None
    per_kernel = {}
    total = 0.0
    for k in kernels:
        f = fetch.get(k, [0.0])[-1] * 1024.0                       # the counters are in KiB
        w = write.get(k, [0.0])[-1] * 1024.0
        per_kernel[k] = {"fetch_bytes": f, "write_bytes": w}
        total += f + w
    g = fetch.get("gather64_kernel")
    calib = (sum(g) / len(g)) * 1024.0 / float(1 << 33) if g else None    # the calibration kernel requests exactly 2^33 B per launch
    return {"bytes_per_launch": total, "per_kernel": per_kernel, "gather64_calibration_ratio": calib,
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate child runs of this script (one step each), last launch of each "
                      "seed-stage kernel; KiB x 1024; not rescaled (FETCH_SIZE reads 1.0x the requested bytes for random 64-B sector gathers, see the ratio)"}


# ------------------------------------------------------------------------------------------------ CPU baseline + parity sample
def write_fastq(codes, n_pairs, fq1, fq2):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    qual = b"I" * codes.shape[1]
    with open(fq1, "wb") as f1, open(fq2, "wb") as f2:
        for p in range(n_pairs):
            f1.write(b"@r%d/1\n" % p + lut[codes[2 * p]].tobytes() + b"\n+\n" + qual + b"\n")
            f2.write(b"@r%d/2\n" % p + lut[codes[2 * p + 1]].tobytes() + b"\n+\n" + qual + b"\n")


def fastq_records(path):
    """annotated FASTQ -> {(name, occurrence): (header, seq, qual)}; record order is not part of the contract."""
    out, seen = {}, {}
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for i in range(0, len(lines) - 3, 4):
        name = lines[i][1:].split(b"\t")[0]
        seen[name] = seen.get(name, 0) + 1
        out[(name, seen[name])] = (lines[i], lines[i + 1], lines[i + 3])
    return out


def cpu_baseline_and_parity(args, arrs, sample_codes, tmpdir, lkt_table, device, read_L, ini_name=None, threads_list=None):
    """Real soap4 (oracle/_ref, built by oracle/Makefile.ref and shipped with the snapshot) on a bounded sample of the bench
    workload, on all host cores; then the same sample through the product executable, compared record by record."""
    from megapath_amd import index_builder
    n_pairs = sample_codes.shape[0] // 2
    fq1, fq2 = os.path.join(tmpdir, "s_1.fq"), os.path.join(tmpdir, "s_2.fq")
    write_fastq(sample_codes.cpu().numpy(), n_pairs, fq1, fq2)
    prefix = os.path.join(tmpdir, "ref.index")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "soap4")
    use_ref = os.path.exists(ref_bin)
    if not os.path.exists(prefix + ".bwt") or (use_ref and not os.path.exists(prefix + ".lkt")):
        index_builder.write_index_files(prefix, arrs, write_lkt=use_ref, lkt_table=lkt_table if use_ref else None)
    cores = os.cpu_count() or 1
    ini_name = ini_name or args.ini
    ini = os.path.join(tmpdir, ini_name)
    with open(ini, "w") as f:
        f.write(INIS[ini_name])
    cb, parity = None, None
    ref_out = os.path.join(tmpdir, "ref.out")
    if use_ref:
        # all host cores, as the north star asks, and a second run at 32 threads (soap4's own thread scaling flattens long before
        # the 256 hardware threads of such a node); the faster one is the baseline, both are listed
        runs = []
        for T in (threads_list or sorted({cores, min(cores, 32)}, reverse=True)):
            cmd = [ref_bin, "pair", prefix, fq1, fq2, "-o", os.path.join(tmpdir, "out"), "-C", ini, "-L", str(read_L), "-T", str(T), "-u", "750", "-F", "-nc"]
            t0 = time.time()
            with open(ref_out + ".T%d" % T, "wb") as fo, open(os.path.join(tmpdir, "ref.err"), "wb") as fe:
                rc = subprocess.call(cmd, stdout=fo, stderr=fe)
            wall = time.time() - t0
            align_s = None
            for line in open(os.path.join(tmpdir, "ref.err"), errors="replace"):
                if "Overall alignment time" in line:
                    align_s = float(line.split(":")[1].split()[0])
            if rc != 0 or not align_s:
                log("[bench] reference soap4 -T %d failed (rc=%d); wall %.1fs" % (T, rc, wall))
                continue
            runs.append({"threads": T, "pairs_per_s": n_pairs / align_s, "alignment_s": align_s, "process_s": wall})
            if not os.path.exists(ref_out):
                os.replace(ref_out + ".T%d" % T, ref_out)
        if runs:
            best = max(runs, key=lambda r: r["pairs_per_s"])
            cb = {"value": best["pairs_per_s"], "unit": "pairs/s", "cores": best["threads"], "host_cores": cores, "kind": "reference", "runs": runs,
                  "sample": "%d pairs of the same workload, real soap4 (oracle/_ref), alignment phase only (index/read load excluded), best of -T %s on "
                            "this box's %d host cores" % (n_pairs, "/".join(str(r["threads"]) for r in runs), cores)}
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        lib = O.oracle()
        ix = lib.mpo_index_open(prefix.encode())
        p = O.default_params(L=read_L)
        p.mmp.seed_min_length, p.mmp.reseed_len = (22, 23) if ini_name == "soap4.ini" else (17, 18)
        st = O.AlignStats()
        few = min(n_pairs, 4000)
        f1s, f2s = os.path.join(tmpdir, "few_1.fq"), os.path.join(tmpdir, "few_2.fq")
        write_fastq(sample_codes.cpu().numpy(), few, f1s, f2s)
        t0 = time.time()
        lib.mpo_align_fastq(ix, C.byref(p), f1s.encode(), f2s.encode(), ref_out.encode(), None, C.byref(st))
        dt = time.time() - t0
        cb = {"value": few / dt, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": "%d pairs of the same workload, scalar C oracle, whole path" % few}
        fq1, fq2, n_pairs = f1s, f2s, few
    # ---- parity at the benchmarked scale: the drop-in executable on the same files, every optional table on
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    if os.path.exists(exe) and os.path.exists(ref_out) and os.path.getsize(ref_out) > 0:
        prod_out = os.path.join(tmpdir, "prod.out")
        env = dict(os.environ)
        env["MP_DEVICE"] = str(device)
        env.pop("LOCAL_RANK", None)
        cmd = [exe, "pair", prefix, fq1, fq2, "-o", os.path.join(tmpdir, "pout"), "-C", ini, "-L", str(read_L), "-T", str(min(cores, 32)), "-u", "750", "-F", "-nc"]
        t0 = time.time()
        with open(prod_out, "wb") as fo, open(os.path.join(tmpdir, "prod.err"), "wb") as fe:
            rc = subprocess.call(cmd, stdout=fo, stderr=fe, env=env)
        if rc != 0:
            parity = {"pairs": n_pairs, "records_equal": False, "first_diff": "product executable failed (rc %d): %s" % (
                rc, open(os.path.join(tmpdir, "prod.err"), errors="replace").read()[-300:])}
        else:
            want, got = fastq_records(ref_out), fastq_records(prod_out)
            diff = [k for k in want if got.get(k) != want[k]] + [k for k in got if k not in want]
            first = None
            if diff:
                k = sorted(diff)[0]
                first = {"record": "%s #%d" % (k[0].decode(), k[1]), "reference": want.get(k, (b"<absent>",))[0].decode(errors="replace")[:300],
                         "product": got.get(k, (b"<absent>",))[0].decode(errors="replace")[:300]}
            aligned = sum(1 for v in want.values() if b"SCORE:0;" not in v[0])
            parity = {"pairs": n_pairs, "records": len(want), "records_equal": not diff, "records_differing": len(diff), "first_diff": first,
                      "reference_records_with_a_hit": aligned, "against": cb["kind"] if cb else None,
                      "how": "bin/soap4-mi355x pair <same index files> <same FASTQ> -C %s -L %d -u 750 -F -nc (whole-process %.0f s), stdout compared with the "
                             "CPU run's, record by record, order-insensitive" % (ini_name, read_L, time.time() - t0)}
    return cb, parity


# ------------------------------------------------------------------------------------------------ the metric's configuration
def nt_replica_leg(args, H, index_builder, simulate, arrs_a, text_a, dev, device, n_bases):
    """BASELINE configs[2] on one GPU: an NT-sized replica of the database resident in HBM in the lean layout (0.92 B/base + a 13-mer
    table per chunk), every batch aligned against every chunk in turn (runMegaPath.sh:191-227 in one process).  Chunk 0 is the genome
    the reads come from; every other chunk is a DIFFERENT genome -- one unrelated synthetic genome of the same size, indexed once and
    copied device-to-device into its own arrays (mp_index_clone) -- so that, as with the real database, a read has its true
    alignment in one chunk and only chance matches in the others.  Read mix of SURVEY 8d C3: 60 % drawn from chunk 0 at 0-8 %
    divergence, 40 % random."""
    knobs = {"MP_KMER_FILTER": os.environ.get("MP_BENCH_NT_KMER", "0"), "MP_KMER15": "0", "MP_LKT16": "0", "MP_SA_DENSE": "0"}
    saved = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    chunks = []
    try:
        t0 = time.time()
        torch.cuda.empty_cache()
        chunks.append(H.Index.from_arrays(device, **arrs_a))
        text_b = simulate.genome(n_bases, seed=4, device=dev)
        tb = text_b.cpu().numpy()
        del text_b
        bounds = np.linspace(0, n_bases, 25).astype(np.int64)
        arrs_b = index_builder.build_index_arrays([("ntB%d" % (i + 1), tb[bounds[i]:bounds[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
        del tb
        torch.cuda.empty_cache()
        chunks.append(H.Index.from_arrays(device, **arrs_b))
        # the real soap4 beside it: the same read mix against ONE of the other chunks (its files written here), on the host cores; a database
        # of n chunks costs it n such runs (runMegaPath.sh:191-227), so its rate for the replica is that of one run / n
        cpu_nt = None
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "soap4")
        if os.path.exists(ref_bin) and not args.no_cpu_baseline:
            td = tempfile.mkdtemp(prefix="mpnt_")
            try:
                prefix = os.path.join(td, "ntB.index")
                index_builder.write_index_files(prefix, arrs_b, write_lkt=True, lkt_table=chunks[1].lkt_table())
                n_cpu = int(os.environ.get("MP_BENCH_NT_CPU_PAIRS", "20000"))
                cw = Workload(text_a, n_cpu, seed=4242, read_len=args.read_len, L=args.read_len + 1, sub_max=0.08, random_frac=0.4)
                f1, f2 = os.path.join(td, "n_1.fq"), os.path.join(td, "n_2.fq")
                write_fastq(cw.codes.cpu().numpy(), n_cpu, f1, f2)
                ini = os.path.join(td, "soap4-nt2.ini")
                with open(ini, "w") as f:
                    f.write(NT2_INI)
                T = min(os.cpu_count() or 1, 32)
                with open(os.devnull, "wb") as fo, open(os.path.join(td, "err"), "wb") as fe:
                    rc = subprocess.call([ref_bin, "pair", prefix, f1, f2, "-o", os.path.join(td, "o"), "-C", ini, "-L", str(args.read_len + 1), "-T", str(T), "-u", "750", "-F", "-nc"],
                                         stdout=fo, stderr=fe)
                al = None
                for line in open(os.path.join(td, "err"), errors="replace"):
                    if "Overall alignment time" in line:
                        al = float(line.split(":")[1].split()[0])
                if rc == 0 and al:
                    cpu_nt = {"kind": "reference", "threads": T, "pairs": n_cpu, "alignment_s_one_chunk": al, "pairs_per_s_one_chunk": n_cpu / al}
            except Exception as e:  # noqa: BLE001
                log("[bench] NT cpu baseline failed: %r" % (e,))
            finally:
                shutil.rmtree(td, ignore_errors=True)
        del arrs_b
        per_chunk = int(chunks[0].info.device_bytes)
        free_b, total_b = torch.cuda.mem_get_info()
        want = int(args.nt_replica_gb * 1e9 // per_chunk) + 1
        fit = 2 + max(0, int((free_b - (20 << 30)) // per_chunk))           # 20 GiB stay free for the batch buffers and traceback slabs
        n_chunks = max(2, min(want, fit))
        t1 = time.time()
        while len(chunks) < n_chunks:
            chunks.append(chunks[1].clone())
        torch.cuda.synchronize()
        t2 = time.time()
        index_bytes = sum(int(x.info.device_bytes) for x in chunks)
        log("[bench] NT replica: %d chunks of %.0f Mbp, %.1f GB of index in HBM (two genomes built in %.1f s, %d device-to-device copies in %.1f s)"
            % (n_chunks, n_bases / 1e6, index_bytes / 1e9, t1 - t0, n_chunks - 2, t2 - t1))
        lp = args.nt_pairs
        wl = Workload(text_a, lp, seed=4242, read_len=args.read_len, L=args.read_len + 1, sub_max=0.08, random_frac=0.4)
        params = H.AlignParams(wl.max_len, 750, 130, 130, H.MmpParams.nt2())
        ctx = H.Context(chunks[0])
        keys = ("seed", "locate", "merge", "pair", "fill", "trace", "rescue_tasks", "rescue_fill", "rescue_trace")

        def one_step(acc):
            ctx.set_reads(wl.packed_np, wl.lens_np)
            for ci, ix in enumerate(chunks):
                ctx.set_index(ix)
                _, nh, st = ctx.deep_dp_round(params, fetch=False)
                _, _, rst = ctx.rescue_round(params, False, fetch=False)
                ctx.pair_hits_view(); ctx.single_hits_view(); ctx.rescue_hits_view()
                if acc is None:
                    continue
                grp = acc["source_chunk" if ci == 0 else "other_chunks"]
                vals = (st.seed.ms_seed, st.seed.ms_locate, st.seed.ms_merge, st.ms_pair, st.ms_dp_fill, st.ms_dp_trace, rst.ms_tasks, rst.ms_dp_fill, rst.ms_dp_trace)
                for k, v in zip(keys, vals):
                    grp[k] = grp.get(k, 0.0) + float(v)
                for k in ("n_occ_loaded", "n_lkt", "n_sa", "n_seedpos", "n_seeds", "n_hits"):
                    grp[k] = grp.get(k, 0) + int(getattr(st.seed, k))
                grp["pair_hits"] = grp.get("pair_hits", 0) + int(nh)
                grp["single_dp_tasks"] = grp.get("single_dp_tasks", 0) + int(rst.n_single_tasks)
                grp["single_hits"] = grp.get("single_hits", 0) + int(rst.n_single_hits)
                grp["chunks"] = grp.get("chunks", 0) + 1
        one_step(None)                                            # first touch of the batch buffers
        torch.cuda.synchronize()
        acc = {"source_chunk": {}, "other_chunks": {}}
        th_a = ctx.dp_thermo_stats()
        t3 = time.perf_counter()
        one_step(acc)
        torch.cuda.synchronize()
        el_single = time.perf_counter() - t3
        th_b = ctx.dp_thermo_stats()
        # how many reads have NO seed of >= 17 bases in a chunk that is not theirs (what a per-chunk pre-screen could skip): the kept seed
        # positions of the batch against chunk 1, per read
        seedless = None
        try:
            ctx.set_reads(wl.packed_np, wl.lens_np)
            ctx.set_index(chunks[1])
            ctx.seed_batch(H.MmpParams.nt2())
            lib = H.load_library()
            off = np.zeros(2 * lp + 1, dtype=np.uint64)
            H._chk(lib.mp_get_seeds(ctx._h, None, 0, off.ctypes.data), "mp_get_seeds")
            per_read = np.diff(off.astype(np.int64))
            per_pair = per_read[0::2] + per_read[1::2]
            seedless = {"reads_without_a_kept_seed_position": float((per_read == 0).mean()), "pairs_without_one": float((per_pair == 0).mean()),
                        "kept_seed_positions_per_read_mean": float(per_read.mean()), "chunk": 1,
                        "note": "seed positions that survive the merge / uniqueness / coverage rules (DV-DPfunctions.cpp:2499-2553) of the batch against a chunk none of "
                                "the reads comes from: a pair without any is the only kind of (pair, chunk) work a pre-screen could skip"}
        except Exception as e:  # noqa: BLE001
            log("[bench] seedless census failed: %r" % (e,))
        ctx.close()
        # the configuration the executable runs in: the batch cut into contiguous blocks, one context (own streams, own host thread) per
        # block, each taking its block through every chunk -- the memory-bound seeding of one overlaps with the VALU-bound DP of the others
        n_ctx = max(1, args.contexts)
        el = el_single
        if n_ctx > 1:
            import threading
            cuts = [lp * i // n_ctx for i in range(n_ctx + 1)]
            sl = [WorkloadSlice(wl, cuts[i], cuts[i + 1]) for i in range(n_ctx)]
            cx = [H.Context(chunks[0]) for _ in range(n_ctx)]
            errs = []

            def through_all(i):
                try:
                    cx[i].set_reads(sl[i].packed_np, sl[i].lens_np)
                    for ix in chunks:
                        cx[i].set_index(ix)
                        cx[i].deep_dp_round(params, fetch=False)
                        cx[i].rescue_round(params, False, fetch=False)
                        cx[i].pair_hits_view(); cx[i].single_hits_view(); cx[i].rescue_hits_view()
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            for rep in range(2):                                      # the first pass touches the new contexts' buffers
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                th = [threading.Thread(target=through_all, args=(i,)) for i in range(n_ctx)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                torch.cuda.synchronize()
                el = time.perf_counter() - t4
            for c in cx:
                c.close()
            if errs:
                raise errs[0]
        o = acc["other_chunks"]
        n_o = max(1, o.get("chunks", 1))
        per_chunk_ms = {k: o.get(k, 0.0) / n_o for k in keys}
        t_seed_s = sum(g.get("seed", 0.0) + g.get("locate", 0.0) for g in acc.values()) / 1e3
        ev = {k: sum(g.get(k, 0) for g in acc.values()) for k in ("n_occ_loaded", "n_lkt", "n_sa", "n_seedpos")}
        seed_bytes = 64 * ev["n_occ_loaded"] + 8 * ev["n_lkt"] + 4 * ev["n_sa"] + 0.25 * wl.sum_len * n_chunks + 16 * ev["n_seedpos"]
        achieved = seed_bytes / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
        return {
            "workload": "%d index chunks of %.0f Mbp resident at once in the lean layout = %.1f Gbp, %.1f GB of HBM (BASELINE configs[2] on one GPU): chunk 0 is the genome "
                        "the reads come from, the other %d hold an unrelated synthetic genome (indexed once, copied device-to-device into separate arrays), so a read "
                        "finds its true alignment in one chunk and chance matches elsewhere, as in the real database; 2x%d bp, 60 %% of the pairs from chunk 0 at 0-8 %% "
                        "divergence, 40 %% random (SURVEY 8d C3); soap4-nt2.ini, -L %d; every batch aligned against each chunk in turn by one context, all three hit "
                        "lists of every chunk fetched" % (n_chunks, n_bases / 1e6, n_chunks * n_bases / 1e9, index_bytes / 1e9, n_chunks - 1, args.read_len, args.read_len + 1),
            "chunks": n_chunks, "bases_resident": n_chunks * n_bases, "index_bytes_in_hbm": index_bytes, "bytes_per_base": index_bytes / (n_chunks * n_bases),
            "hbm_total_bytes": int(total_b), "load_s": t2 - t0, "pairs_per_step": lp, "ms_per_step": el * 1e3, "pairs_per_s": lp / el, "contexts": n_ctx,
            "single_context": {"ms_per_step": el_single * 1e3, "pairs_per_s": lp / el_single, "note": "the step the per-chunk kernel times and the roofline below come from"},
            "chunk_pair_alignments_per_s": lp * n_chunks / el,
            "cpu_baseline": (dict(cpu_nt, value=cpu_nt["pairs_per_s_one_chunk"] / n_chunks, unit="pairs/s",
                                  sample="%d pairs of the same read mix, real soap4 (oracle/_ref) -T %d against ONE chunk (the unrelated genome); the replica's %d "
                                         "chunks are %d such runs for it, so value = that rate / %d" % (cpu_nt["pairs"], cpu_nt["threads"], n_chunks, n_chunks, n_chunks))
                             if cpu_nt else None),
            "kernel_ms_per_chunk_other_chunks": per_chunk_ms, "kernel_ms_source_chunk": {k: acc["source_chunk"].get(k, 0.0) for k in keys},
            "foreign_chunk_seed_census": seedless,
            "rescue_certificate_per_pair_per_chunk": dict(zip(("certified_no_alignment", "score_of_%d_elsewhere" % 10, "bound_reaches_cutoff", "not_eligible"),
                                                              [(b - a) / (lp * n_chunks) for a, b in zip(th_a, th_b)]),
                                                          note="dp_thermo_kernel (kernels_thermo.hip) over the single-end rescue instances of the one-context step: "
                                                               "only what it does not certify goes through the score-only filter"),
            "per_pair_per_other_chunk": {k: o.get(k, 0) / (n_o * lp) for k in ("n_occ_loaded", "n_lkt", "n_sa", "n_seeds", "n_hits", "n_seedpos", "single_dp_tasks", "single_hits", "pair_hits")},
            "roofline": {"bound": "hbm", "kernel": "seed_kernel+seed_overflow_kernel+expand_hits_kernel+locate_kernel (lean layout, all chunks of the step)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_pair_per_chunk": seed_bytes / (lp * n_chunks), "launch_ms_per_chunk": t_seed_s * 1e3 / n_chunks,
                         "formula": "64*rank_blocks_loaded + 8*lkt13_lookups + 4*sa_reads + 0.25*read_bases + 16*seed_positions_out over the seed-stage kernels' time"},
        }
    finally:
        for x in chunks:
            try:
                x.close()
            except Exception:  # noqa: BLE001
                pass
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ------------------------------------------------------------------------------------------------ through the boundary
def write_fastq_fast(codes_dev, path1, path2, first_id=0):
    """2P x L base codes on the device -> two FASTQ files with fixed-width names, formatted as one byte matrix per file on the
    device (a Python loop over millions of records would take longer than the run it feeds)."""
    n2, L = codes_dev.shape
    dev = codes_dev.device
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ids = torch.arange(first_id, first_id + n2 // 2, device=dev, dtype=torch.int64)
    digits = torch.stack([(ids // (10 ** k)) % 10 for k in range(8, -1, -1)], dim=1).to(torch.uint8) + 48       # 9 digits
    for m, path in ((0, path1), (1, path2)):
        P = n2 // 2
        rec = torch.empty(P, 2 + 9 + 3 + L + 3 + L + 1, dtype=torch.uint8, device=dev)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:11] = digits
        rec[:, 11] = ord("/"); rec[:, 12] = ord("1") + m; rec[:, 13] = 10
        rec[:, 14:14 + L] = lut[codes_dev[m::2].long()]
        o = 14 + L
        rec[:, o] = 10; rec[:, o + 1] = ord("+"); rec[:, o + 2] = 10
        # qualities as a current sequencer writes them: four bins (F : , #), mostly the top one -- a constant quality string would make the
        # gzip'ed variant of the files compress (and inflate) unrealistically well
        u = torch.rand(P, L, device=dev)
        q = torch.full((P, L), ord("F"), dtype=torch.uint8, device=dev)
        q[u > 0.90] = ord(":"); q[u > 0.96] = ord(","); q[u > 0.99] = ord("#")
        rec[:, o + 3:o + 3 + L] = q
        del u, q
        rec[:, o + 3 + L] = 10
        with open(path, "ab") as f:
            f.write(rec.cpu().numpy().tobytes())
        del rec


def _bgzf_block(raw):
    import struct
    import zlib
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = c.compress(raw) + c.flush()
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(raw), len(raw))


def write_bgzf(plain_path, dst, n_bytes, block=65280):
    """The first n_bytes of a file as BGZF: 64 KiB members with the 'BC' extra field and the empty end-of-file member, compressed by a pool of processes."""
    import multiprocessing as mp
    left = n_bytes
    with open(plain_path, "rb") as f, open(dst, "wb") as fo, mp.Pool(min(16, os.cpu_count() or 4)) as pool:
        while left > 0:
            data = f.read(min(block * 512, left))
            if not data:
                break
            left -= len(data)
            for blk in pool.imap(_bgzf_block, [data[i:i + block] for i in range(0, len(data), block)], chunksize=16):
                fo.write(blk)
        fo.write(_bgzf_block(b""))


def e2e_cli_leg(args, text, prefix, ini_path, read_L, device, tmpdir, n_batches=12, gz_pairs=1 << 23):
    """FASTQ files in, annotated FASTQ out, through bin/soap4-mi355x (the boundary a pipeline maintainer sees): pairs/s of the
    executable's alignment loop (index load excluded, as for the CPU soap4), stdout to /dev/null; plain input and gzip'ed input."""
    from megapath_amd import simulate
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    if not os.path.exists(exe) or not os.path.exists(prefix + ".bwt"):
        return {"error": "executable or index files missing"}
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tmpdir
    d = tempfile.mkdtemp(prefix="mpe2e_", dir=shm)
    res = {}
    try:
        P = args.pairs_per_step
        f1, f2 = os.path.join(d, "r_1.fq"), os.path.join(d, "r_2.fq")
        t0 = time.time()
        for b in range(n_batches):
            codes, _ = simulate.paired_reads(text, P, args.read_len, seed=7000 + b)
            write_fastq_fast(codes, f1, f2, first_id=b * P)
            del codes
        torch.cuda.empty_cache()
        t_gen = time.time() - t0
        threads = min(os.cpu_count() or 16, int(os.environ.get("MP_BENCH_E2E_THREADS", "16")))

        def run(a, b, n_pairs, extra_env=None):
            env = dict(os.environ, MP_TIMING="1", MP_DEVICE=str(device))
            env.update(extra_env or {})
            env.pop("LOCAL_RANK", None)
            cmd = [exe, "pair", prefix, a, b, "-o", os.path.join(d, "o"), "-C", ini_path, "-L", str(read_L), "-T", str(threads), "-u", "750", "-F", "-nc"]
            t0 = time.time()
            with open(os.devnull, "wb") as fo:
                r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
            wall = time.time() - t0
            err = r.stderr.decode(errors="replace")
            loop_s = None
            for line in err.splitlines():
                if "alignment loop" in line:
                    loop_s = float(line.split("alignment loop")[1].split()[0])
            rd = [l for l in err.splitlines() if "reader:" in l and " 0 pairs" not in l]
            rd_t = [(float(l.split("lines found in")[1].split()[0]), float(l.split("packed in")[1].split()[0])) for l in rd[len(rd) // 2:]]     # second half: steady state
            out = {"rc": r.returncode, "pairs": n_pairs, "process_s": wall, "alignment_loop_s": loop_s, "pairs_per_s": (n_pairs / loop_s) if loop_s else None,
                   "threads": threads, "stage_seconds": next((l.split("seconds:")[1].strip() for l in err.splitlines() if "seconds:" in l), None),
                   "reader_ms_per_batch_steady": {"lines": 1e3 * sum(a for a, _ in rd_t) / len(rd_t), "records_and_packing": 1e3 * sum(b for _, b in rd_t) / len(rd_t)} if rd_t else None}
            if r.returncode != 0:
                out["stderr_tail"] = err[-400:]
            return out
        res["plain_first_run"] = run(f1, f2, n_batches * P)       # the first process after the files were written runs slower (cold pages, clocks): listed, not quoted
        res["plain"] = run(f1, f2, n_batches * P)
        # tuning aid: MP_BENCH_E2E_VARIANTS="name:ENV=val,ENV=val;name2:..." runs the plain input again under each environment
        for spec in filter(None, os.environ.get("MP_BENCH_E2E_VARIANTS", "").split(";")):
            nm, _, kv = spec.partition(":")
            res["plain_" + nm] = run(f1, f2, n_batches * P, dict(x.split("=", 1) for x in kv.split(",") if "=" in x))
        res["plain"]["input"] = "%d pairs, 2 x %.2f GB of FASTQ in %s (generated in %.0f s)" % (n_batches * P, os.path.getsize(f1) / 1e9, shm, t_gen)
        # the NT loop in one process (review item 7: is the per-chunk merge of the header comments on the host in the way?): the same index
        # given eight times in the lean layout, NT read mix; the printer merges every batch's comments chunk after chunk
        try:
            n1, n2 = os.path.join(d, "nt_1.fq"), os.path.join(d, "nt_2.fq")
            ntp = min(P, 1 << 19)
            codes, _ = simulate.paired_reads(text, ntp, args.read_len, seed=7100, sub_max=0.08, random_frac=0.4)
            write_fastq_fast(codes, n1, n2)
            del codes
            nt_ini = os.path.join(d, "soap4-nt2.ini")
            with open(nt_ini, "w") as f:
                f.write(NT2_INI)
            env = dict(os.environ, MP_TIMING="1", MP_DEVICE=str(device), MP_KMER_FILTER="0", MP_KMER15="0", MP_LKT16="0", MP_SA_DENSE="0")
            env.pop("LOCAL_RANK", None)
            rc8 = {}
            for k in (2, 8):
                for label, extra in (("device_reduce", {}), ("host_merge", {"MP_HOST_MERGE": "1"})):
                    cmd = [exe, "pair", ",".join([prefix] * k), n1, n2, "-o", os.path.join(d, "o"), "-C", nt_ini, "-L", str(args.read_len + 1), "-T", str(threads), "-u", "750", "-F", "-nc", "-top", "95"]
                    with open(os.devnull, "wb") as fo:
                        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=dict(env, **extra))
                    err = r.stderr.decode(errors="replace")
                    loop_s = next((float(l.split("alignment loop")[1].split()[0]) for l in err.splitlines() if "alignment loop" in l), None)
                    rc8["%d_chunks_%s" % (k, label)] = {"rc": r.returncode, "alignment_loop_s": loop_s, "pairs_per_s": ntp / loop_s if loop_s else None,
                                                        "stage_seconds": next((l.split("seconds:")[1].strip() for l in err.splitlines() if "seconds:" in l), None)}
            rc8["input"] = ("%d pairs of the NT read mix (60 %% at 0-8 %% divergence, 40 %% random) against the same lean index given 2 and 8 times; device_reduce = the "
                            "per-reference scores reduced on the device over all chunks, ONE printer pass per batch (mp_reduce_*); host_merge = hit lists to the host, header "
                            "comments merged chunk after chunk (MP_HOST_MERGE=1, the path -b takes).  'output of ...' in stage_seconds is the printer's time" % ntp)
            res["resident_chunks"] = rc8
        except Exception as e:  # noqa: BLE001
            res["resident_chunks"] = {"error": repr(e)}
        # gzip'ed input (what runMegaPath.sh:136 feeds): one zlib stream per file inflates at a few hundred MB/s
        g1, g2 = os.path.join(d, "g_1.fq"), os.path.join(d, "g_2.fq")
        rec_bytes = os.path.getsize(f1) // (n_batches * P)
        for src, dst in ((f1, g1), (f2, g2)):
            with open(src, "rb") as fi, open(dst, "wb") as fo:
                fo.write(fi.read(rec_bytes * gz_pairs))
        t0 = time.time()
        procs = [subprocess.Popen(["gzip", "-1", g]) for g in (g1, g2)]
        for pr in procs:
            pr.wait()
        res["gz"] = run(g1 + ".gz", g2 + ".gz", gz_pairs)                                       # several threads per stream (host/par_gzip.h)
        res["gz_one_thread_per_file"] = run(g1 + ".gz", g2 + ".gz", gz_pairs, {"MP_PARGZ_MIN": str(1 << 60)})     # the reader's own decoder, one thread per stream
        res["gz_zlib"] = run(g1 + ".gz", g2 + ".gz", gz_pairs, {"MP_INFLATE": "zlib"})        # the same files through zlib's gzread, as the reference reads them
        for spec in filter(None, os.environ.get("MP_BENCH_GZ_VARIANTS", "").split(";")):      # tuning aid, as MP_BENCH_E2E_VARIANTS
            nm, _, kv = spec.partition(":")
            res["gz_" + nm] = run(g1 + ".gz", g2 + ".gz", gz_pairs, dict(x.split("=", 1) for x in kv.split(",") if "=" in x))
        # the same reads as BGZF (bgzip; what BBTools writes when bgzip is installed): independent members, inflated by several threads per file
        try:
            t0 = time.time()
            b1, b2 = os.path.join(d, "b_1.fq.gz"), os.path.join(d, "b_2.fq.gz")
            # (in child processes of their own: the compression pool forks, and this process has the GPU open)
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--write-bgzf", src, dst, str(rec_bytes * gz_pairs)]) for src, dst in ((f1, b1), (f2, b2))]
            if any(pr.wait() != 0 for pr in ps):
                raise RuntimeError("writing the BGZF files failed")
            res["bgzf"] = run(b1, b2, gz_pairs)
            res["bgzf"]["input"] = "%d pairs as BGZF (%.2f GB per file, written in %.0f s)" % (gz_pairs, os.path.getsize(b1) / 1e9, time.time() - t0)
        except Exception as e:  # noqa: BLE001
            res["bgzf"] = {"error": repr(e)}
        res["gz"]["input"] = "%d pairs, gzip -1 (%.2f GB per file, compressed in %.0f s)" % (gz_pairs, os.path.getsize(g1 + ".gz") / 1e9, time.time() - t0)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    res["note"] = ("pairs/s of the executable's alignment loop: reading, packing, device stage, grouping, formatting and writing (to /dev/null) of every "
                   "record; the index load is outside, as it is for the CPU soap4's 'Overall alignment time'")
    return res


# ------------------------------------------------------------------------------------------------ N ranks from one command line
def launch_ranks(n, argv, script=None, extra_env=None):
    """`python bench.py --gpus N` started without a launcher (no RANK in the environment): start N fresh child processes of
    this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set
    them), wait for all of them and return the worst exit code.  Rank 0 inherits stdout, so its ONE JSON line is this process's
    JSON line; the other ranks' stdout goes to stderr.  The parent never touches the GPU and never execs: the children are
    ordinary subprocesses.  A rank that dies takes the others down (they would wait in the barrier forever)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": env.get("MASTER_PORT", str(port)),
                    "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is relayed line by line: JSON lines go to stdout, anything else a library prints there (gloo's connection
    # banner) goes to stderr, so that stdout stays ONE JSON line
    import threading

    def relay(pipe):
        for raw in iter(pipe.readline, b""):
            line = raw.decode(errors="replace")
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush(); sys.stderr.flush()
    pump = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    pump.start()
    codes = {}
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            codes[r] = code
            if code != 0:
                log("[bench] rank %d exited with code %d" % (r, code))
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    pump.join(timeout=10)
    bad = [c for c in codes.values() if c != 0]
    return 0 if not bad else max([c for c in bad if c > 0] or [1])       # the worst exit code; a rank killed by a signal counts as 1


# ------------------------------------------------------------------------------------------------ main
def main():
    if sys.argv[1:2] == ["--write-bgzf"]:                       # helper mode of the e2e leg (no GPU is touched)
        write_bgzf(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("MP_BENCH_GENOME_MBP", "2950")))
    ap.add_argument("--pairs-per-step", type=int, default=int(os.environ.get("MP_BENCH_PAIRS", str(1 << 20))))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ini", choices=sorted(INIS), default=os.environ.get("MP_BENCH_INI", "soap4.ini"),
                    help="soap4.ini + -L <read-len> is how runMegaPath.sh:136 runs the human-filter stage (configs[1]); soap4-nt2.ini + -L <read-len + 1> is the NT stage's")
    ap.add_argument("--L", type=int, default=None, help="-L value (maxReadLength; reads are cut to L - 1); default: read-len for soap4.ini, read-len + 1 for soap4-nt2.ini")
    ap.add_argument("--cpu-sample-pairs", type=int, default=int(os.environ.get("MP_BENCH_CPU_PAIRS", "100000")))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the parity sample")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 child runs (roofline.traffic = null)")
    ap.add_argument("--no-legs", action="store_true", help="skip the C3/C4/C5 workload legs, the lean-layout leg and the resident-chunks leg")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("MP_BENCH_CONTEXTS", "3")),
                    help="contexts (HIP stream pairs + host threads) a step's batch is cut across; 1 = one context does the whole batch")
    ap.add_argument("--nt-replica-gb", type=float, default=float(os.environ.get("MP_BENCH_NT_GB", "242")),
                    help="HBM bytes (GB) of index the nt_replica leg makes resident (as many lean chunks of --genome-mbp as that takes; 0 = skip the leg)")
    ap.add_argument("--nt-pairs", type=int, default=int(os.environ.get("MP_BENCH_NT_PAIRS", "65536")), help="pairs per step of the nt_replica leg")
    ap.add_argument("--legs", default=os.environ.get("MP_BENCH_LEGS", "all"), help="comma-separated legs to run (e2e_cli, nt2_ini_same_reads, C3_divergent, C4_2x250, "
                    "C5_mixed_rescue, lean_layout, nt_replica) or 'all'")
    ap.add_argument("--no-e2e", action="store_true", help="skip the leg through the executable (FASTQ files in, annotated FASTQ out)")
    ap.add_argument("--e2e-batches", type=int, default=int(os.environ.get("MP_BENCH_E2E_BATCHES", "24")), help="batches of --pairs-per-step pairs in the e2e leg's input files")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    leg_set = None if args.legs == "all" else set(x for x in args.legs.split(",") if x)

    def want_leg(name):
        return leg_set is None or name in leg_set
    if args.pmc_child:
        args.no_cpu_baseline = args.no_pmc = args.no_legs = True

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: be the launcher (nothing here has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knobs (1-GPU boxes): MP_BENCH_SHARE_GPU=1 puts every rank on device 0, MP_BENCH_BACKEND=gloo avoids RCCL
    if os.environ.get("MP_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()                 # what the process group really spans: this is what n_gpus reports
        if world != args.gpus:
            log("[bench] --gpus %d but the process group has %d ranks; reporting %d" % (args.gpus, world, world))
    barrier = dist.barrier if world > 1 else None

    from megapath_amd import hipapi as H
    from megapath_amd import index_builder, simulate, taxcount

    n_bases = int(args.genome_mbp * 1e6)
    t0 = time.time()
    text = simulate.genome(n_bases, seed=3, device=dev)
    n_seqs = 24
    bounds = np.linspace(0, n_bases, n_seqs + 1).astype(np.int64)
    text_cpu = text.cpu().numpy()
    seqs = [("chr%d" % (i + 1), text_cpu[bounds[i]:bounds[i + 1]]) for i in range(n_seqs)]
    arrs = index_builder.build_index_arrays(seqs, sa_interval=16, device=dev)
    t1 = time.time()
    torch.cuda.empty_cache()                       # the suffix sort's cached blocks would hide the free HBM from the index loader
    gix = H.Index.from_arrays(local_rank, **arrs)
    t2 = time.time()
    if rank == 0:
        bi = index_builder.last_build_info or {}
        log("[bench] genome %.0f Mbp: generated + index built in %.1fs (mp_build_index_arrays: suffix array in %.2f s, %s doubling rounds), HBM re-layout + upload %.1fs, "
            "index %.2f GB in HBM" % (args.genome_mbp, t1 - t0, bi.get("ms_suffix_sort", 0.0) / 1e3, bi.get("rounds", "?"), t2 - t1, gix.info.device_bytes / 1e9))
    torch.cuda.empty_cache()
    g_ms, g_bytes = gix.bench_gather64()
    gather_gbs = g_bytes / (g_ms * 1e-3) / 1e9
    if rank == 0:
        log("[bench] random 64-B gather ceiling over the rank blocks: %.0f GB/s" % gather_gbs)

    P = args.pairs_per_step
    read_L = args.L if args.L is not None else (args.read_len if args.ini == "soap4.ini" else args.read_len + 1)
    wl = Workload(text, P, seed=1000 + rank, read_len=args.read_len, L=read_L)
    ctx = H.Context(gix)
    # insert_low = the longest read of the first batch (SOAP4.cpp:465-474), as the executable settles it
    params = H.AlignParams(wl.max_len, 750, 130, 130, mmp_of(args.ini))

    for _ in range(max(args.warmup, 0 if args.pmc_child else 1)):
        step(ctx, wl, params)
    census = ref_events = None
    if not args.pmc_child:
        # two untimed steps over the same resident batch: (1) the cells every DP launch computes, counted on the device;
        # (2) the look-ups of the reference algorithm itself -- every optional table ignored -- for `equivalent_work`
        ctx.set_dp_census(True)
        step(ctx, wl, params)
        census = ctx.dp_census()
        ctx.set_dp_census(False)
        if gix.info.prefilter_k:
            ctx.set_prefilter(False)
            _, _, st0, _ = step(ctx, wl, params)
            ref_events = dict(occ=int(st0.seed.n_occ), lkt=int(st0.seed.n_lkt), sa=int(st0.seed.n_sa), seedpos=int(st0.seed.n_seedpos))
            ctx.set_prefilter(True)
    band0, th0 = ctx.dp_band_stats(), ctx.dp_thermo_stats()
    elapsed_single, acc = run_steps(ctx, wl, params, args.steps, barrier)
    band1, th1 = ctx.dp_band_stats(), ctx.dp_thermo_stats()
    elapsed = elapsed_single
    n_ctx = max(1, args.contexts) if not args.pmc_child else 1
    if n_ctx > 1:
        # the configuration the drop-in executable runs in (MP_WORKERS_PER_GPU): the batch cut into contiguous blocks, one context each
        cuts = [P * i // n_ctx for i in range(n_ctx + 1)]
        slices = [WorkloadSlice(wl, cuts[i], cuts[i + 1]) for i in range(n_ctx)]
        ctxs = [ctx] + [H.Context(gix) for _ in range(n_ctx - 1)]
        run_overlapped(ctxs, slices, params, 1)                      # first touch of the new contexts' buffers
        elapsed = run_overlapped(ctxs, slices, params, args.steps, barrier)
        for c in ctxs[1:]:
            c.close()
        step(ctx, wl, params)                                        # leave the whole batch's hit lists in ctx for the exchange step below
    # the exchange step of the path: per-reference hit counts of this rank's last step, summed over the ranks
    hits, nh = ctx.pair_hits_view()
    ref_counts = taxcount.count_pair_hits_per_reference(gix, hits, nh)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ref_counts = taxcount.allreduce_counts(ref_counts, dev if backend == "nccl" else "cpu")

    if rank == 0:
        K = args.steps
        total_pairs = world * K * P
        t_seed_s = (acc["ms_seed"] + acc["ms_locate"]) / 1e3
        # whole suffix array resident: expand_hits_kernel / locate_kernel are not launched (merge_kernel<true> makes its keys from the seed records)
        fused = acc["ms_locate"] / K < 0.05
        roof_kernels = ("seed_kernel", "seed_overflow_kernel") if fused else SEED_STAGE_KERNELS
        b_loaded = loaded_bytes(acc, wl.sum_len, K, fused)
        achieved = b_loaded / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
        kernel_ms = {k[3:]: acc[k] / K for k in acc if k.startswith("ms_")}
        per_pair = lambda k: acc[k] / (K * P)      # noqa: E731
        out = {
            "metric": "paired 150bp reads aligned/sec vs 2BWT index",
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "contexts_per_gpu": n_ctx,
            "single_context": {"pairs_per_s": world * K * P / elapsed_single, "ms_per_step": elapsed_single / K * 1e3,
                               "note": "the same %d steps with ONE context doing the whole batch on one stream: the run the per-kernel HIP-event times, "
                                       "event counts and the roofline below come from (kernels of several contexts would overlap and stretch each other)" % K},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "2x%dbp synthetic pairs vs synthetic %.0f Mbp genome in 24 sequences (BASELINE configs[1], human-filter stage; GRCh38 indexes ~2.95 Gbp of "
                                   "non-N bases), %d pairs/step/GPU; a step = upload of the packed reads + deep-DP round (MMP seeding, pairing, both DP legs + traceback) + "
                                   "rescue round (single-end DP, mate rescue) + all three hit lists back in pinned host memory; the batch is cut into %d contiguous blocks "
                                   "that as many contexts work on side by side" % (args.read_len, args.genome_mbp, P, n_ctx),
                       "index_replicated_per_gpu": True, "pairs_per_step_per_gpu": P, "genome_mbp": args.genome_mbp,
                       "index_bytes_in_hbm": int(gix.info.device_bytes), "ini": args.ini, "L": read_L, "u": 750,
                       "command_line": "soap4 pair <index> r1.fq r2.fq -C %s -L %d -u 750 -F -nc (runMegaPath.sh:136)" % (args.ini, read_L)},
            "roofline": {"bound": "hbm", "kernel": "+".join(roof_kernels), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": b_loaded / K, "bytes_per_pair": b_loaded / (K * P), "launch_ms": (acc["ms_seed"] + acc["ms_locate"]) / K,
                         "formula": ("64*rank_blocks_loaded + 8*(lkt13 + lkt16 look-ups) + 8*prefilter_entries_loaded + 4*sa_reads%s + 8*text_words_compared + "
                                     "0.25*read_bases + 16*seed_positions_out; every count is emitted by the kernels (events_per_pair) and only events "
                                     "that load memory are priced") % (" (the seeding kernel's own, %.2f per pair: the kernels in the denominator)" % (acc["n_sa_seed_kernel"] / (K * P)) if fused else ""),
                         "random64_gather_ceiling_gbs": gather_gbs,
                         "note": "dependent random 4..64-byte gathers: the physical ceiling of this access shape is the measured random-64-B-gather rate, "
                                 "not the streaming peak; traffic_frac / frac_of_gather_ceiling say how close the sectors actually moved come to either"},
            "kernel_ms_per_step": kernel_ms,
            "events_per_pair": {"rank_queries_issued": per_pair("n_occ"), "rank_blocks_loaded": per_pair("n_occ_loaded"), "lkt13_lookups": per_pair("n_lkt"),
                                "lkt16_lookups": per_pair("n_lkt16"), "sa_reads": per_pair("n_sa"), "prefilter_probes": per_pair("n_prefilter"),
                                "prefilter_entries_loaded": per_pair("n_prefilter_loads"), "text_words_compared": per_pair("n_text_words"),
                                "extension_steps_skipped": per_pair("n_steps_skipped"), "seeds": per_pair("n_seeds"), "located_hits": per_pair("n_hits"),
                                "seed_positions_out": per_pair("n_seedpos"), "candidates": per_pair("cand"), "pair_hits": per_pair("pair_hits"),
                                "failed_pairs": per_pair("failed"), "single_dp_tasks": per_pair("stasks"), "rescue_dp_slots": per_pair("rtasks"),
                                "single_hits": per_pair("single_hits"), "rescue_hits": per_pair("rescue_hits")},
            "hit_count_allreduce": {"references": int(len(ref_counts)), "pair_hits_counted": int(ref_counts.sum()), "ranks": world,
                                    "backend": (backend if world > 1 else None)},
        }
        # ---- what the short cuts stand for (NOT in frac): the same walk priced as if every step had issued its rank queries,
        # and the reference algorithm's own look-ups (SURVEY 8d formula, reference layouts: 64 B/occ, 16 B/lkt, 8 B/sa)
        eq_occ = acc["n_occ"] + 2 * acc["n_steps_skipped"]
        eq = {"rank_queries_incl_skipped_steps_per_pair": eq_occ / (K * P),
              "note": "extension steps the 16-mer table and the in-text comparison stand for, counted as 2 rank queries each; no memory is touched for them"}
        if ref_events:
            b_ref = 64 * ref_events["occ"] + 16 * ref_events["lkt"] + 8 * ref_events["sa"] + 0.25 * wl.sum_len + 16 * ref_events["seedpos"]
            eq["reference_algorithm"] = {"bytes_per_pair": b_ref / P, "events_per_pair": {k: v / P for k, v in ref_events.items()},
                                         "delivered_gbs": b_ref * K / t_seed_s / 1e9 if t_seed_s > 0 else 0.0,
                                         "note": "every table look-up and rank query soap4 itself issues for this batch (one untimed step with the optional "
                                                 "tables ignored), SURVEY 8d prices; 'delivered' = those bytes over the timed kernels' duration -- a work-"
                                                 "equivalent rate, not a bandwidth"}
        out["equivalent_work"] = eq
        # ---- DP: cells the kernels really compute (census step) over the DP fill time
        t_dp_s = (acc["ms_fill"] + acc["ms_rescue_fill"]) / 1e3 / K
        if census is not None and t_dp_s > 0:
            computed = census.cells_band_score + census.cells_band_trace + census.cells_full_score + census.cells_full_trace
            out["dp"] = {"bound": "valu", "unit": "Gcells/s", "achieved": computed / t_dp_s / 1e9,
                         "cells_computed_per_step": {"band_score_pass": int(census.cells_band_score), "band_traceback_pass": int(census.cells_band_trace),
                                                     "full_table_score_only": int(census.cells_full_score), "full_table_with_traceback": int(census.cells_full_trace)},
                         "instances_per_step": int(census.instances),
                         "full_table_equivalent": {"cells_per_step": int(census.cells_full_table), "gcells_per_s": census.cells_full_table / t_dp_s / 1e9,
                                                   "note": "(window+1) x read length of every instance submitted (SURVEY 8d; what callDP computes): a work-equivalent "
                                                           "rate, the kernels skip most of these cells"},
                         "band_instances_per_step": (band1[0] - band0[0]) / K, "band_left_to_full_table_per_step": (band1[1] - band0[1]) / K,
                         "rescue_certificate_per_step": dict(zip(("certified_no_alignment", "score_of_10_elsewhere", "bound_reaches_cutoff", "not_eligible"),
                                                                 [(b - a) / K for a, b in zip(th0, th1)])),
                         "fill_ms_per_step": t_dp_s * 1e3,
                         "note": "integer max/add recurrences: VALU-bound, no MFMA, not an HBM roofline (SURVEY 8d).  Cells counted on the device by "
                                 "mp_ctx_set_dp_census in one untimed step of the same batch; empty mate-rescue slots and instances the block-match "
                                 "pre-filter finishes as exact matches compute no cells.  VALU occupancy per kernel: profiles/ (SQ counter pass)"}
        tmpdir = tempfile.mkdtemp(prefix="mpbench_")
        try:
            if world == 1 and not args.no_pmc:
                try:
                    tr = live_traffic(args, tmpdir, roof_kernels)
                except Exception as e:  # noqa: BLE001
                    log("[bench] PMC passes failed: %r" % (e,))
                    tr = None
                if tr:
                    rf = out["roofline"]
                    rf["traffic"] = tr["bytes_per_launch"]
                    t_launch = rf["launch_ms"] / 1e3
                    rf["traffic_gbs"] = tr["bytes_per_launch"] / t_launch / 1e9
                    rf["traffic_frac"] = rf["traffic_gbs"] / HBM_PEAK_GBS
                    rf["frac_of_gather_ceiling"] = rf["traffic_gbs"] / gather_gbs
                    rf["traffic_over_requested"] = tr["bytes_per_launch"] / rf["bytes_per_launch"]
                    rf["traffic_detail"] = tr
            cb = parity = None
            if world == 1 and not args.no_cpu_baseline:
                try:
                    cb, parity = cpu_baseline_and_parity(args, arrs, wl.codes[: 2 * min(args.cpu_sample_pairs, P)], tmpdir, gix.lkt_table(), local_rank, read_L)
                except Exception as e:  # noqa: BLE001 -- the baseline is auxiliary; report its absence, never hide the GPU number
                    log("[bench] cpu baseline / parity sample failed: %r" % (e,))
            out["cpu_baseline"] = cb
            out["parity_sample"] = parity
            if world == 1 and not args.no_cpu_baseline and args.ini != "soap4-nt2.ini" and cb and cb.get("kind") == "reference":
                # the NT stage's parameters on the same index (17-base seeds: the single-end rescue over chance seeds is exercised): a second sample
                try:
                    os.remove(os.path.join(tmpdir, "ref.out")) if os.path.exists(os.path.join(tmpdir, "ref.out")) else None
                    nt_sample = Workload(text, min(args.cpu_sample_pairs, P) // 2, seed=1000 + rank, read_len=args.read_len, L=args.read_len + 1)
                    cb2, parity2 = cpu_baseline_and_parity(args, arrs, nt_sample.codes, tmpdir, None, local_rank, args.read_len + 1, ini_name="soap4-nt2.ini", threads_list=[min(os.cpu_count() or 1, 32)])
                    out["parity_sample_nt2_ini"] = dict(parity2 or {}, cpu_pairs_per_s=(cb2 or {}).get("value"))
                    del nt_sample
                except Exception as e:  # noqa: BLE001
                    log("[bench] nt2 parity sample failed: %r" % (e,))
            if world == 1 and not args.no_legs and not args.no_e2e and want_leg("e2e_cli"):
                try:
                    from megapath_amd import index_builder as _ib
                    prefix = os.path.join(tmpdir, "ref.index")
                    if not os.path.exists(prefix + ".bwt"):
                        _ib.write_index_files(prefix, arrs, write_lkt=False)
                    ini_path = os.path.join(tmpdir, args.ini)
                    with open(ini_path, "w") as f:
                        f.write(INIS[args.ini])
                    out["e2e_cli"] = e2e_cli_leg(args, text, prefix, ini_path, read_L, local_rank, tmpdir, n_batches=args.e2e_batches)
                except Exception as e:  # noqa: BLE001
                    out["e2e_cli"] = {"error": repr(e)}
                    log("[bench] e2e leg failed: %r" % (e,))
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)
        # ---- workload legs: the other configs' read shapes against the same index (sensitivity of the fast paths)
        if world == 1 and not args.no_legs:
            legs = {}
            specs = [
                ("C3_divergent", "2x150 bp, per-pair divergence uniform 0-8 %, 40 % random pairs (NT-stage read mix)", dict(read_len=150, sub_max=0.08, random_frac=0.4)),
                ("C4_2x250", "2x250 bp, -L 251, fragments ~N(450,50): 311-row windows, 32-lane DP shapes", dict(read_len=250, frag_mean=450.0, frag_sd=50.0, frag_max=700)),
                ("C5_mixed_rescue", "lengths uniform 75-300 (-L 301), 30 % of pairs with one mate random: rescue stress", dict(
                    read_len=300, len_range=(75, 300), broken_mate_frac=0.3, frag_mean=450.0, frag_sd=50.0, frag_max=700)),
            ]
            # every leg below is an NT-stage shape: soap4-nt2.ini (17-base seeds), -L = read length + 1 (runMegaPath.sh:206)
            nt_L = args.read_len + 1
            nt_wl = wl if (args.ini == "soap4-nt2.ini" and read_L == nt_L) else Workload(text, max(P // 4, 1 << 16), seed=1000 + rank, read_len=args.read_len, L=nt_L)
            nt_params = H.AlignParams(nt_wl.max_len, 750, 130, 130, H.MmpParams.nt2())
            try:
                if not want_leg("nt2_ini_same_reads"):
                    raise KeyError("skipped")
                lp = nt_wl.n_pairs
                lctx = H.Context(gix)
                step(lctx, nt_wl, nt_params)
                el, la = run_steps(lctx, nt_wl, nt_params, 2)
                legs["nt2_ini_same_reads"] = {"workload": "the default read mix with the NT stage's parameters (-C soap4-nt2.ini -L %d: 17-base seeds, reads of %d bases) against the same "
                                                          "index, one context: the configuration rounds 1 and 2 quoted" % (nt_L, nt_wl.max_len),
                                              "pairs_per_step": lp, "pairs_per_s": 2 * lp / el, "ms_per_step": el / 2 * 1e3,
                                              "kernel_ms_per_step": {k[3:]: la[k] / 2 for k in la if k.startswith("ms_")}}
                lctx.close()
            except KeyError:
                pass
            except Exception as e:  # noqa: BLE001
                legs["nt2_ini_same_reads"] = {"error": repr(e)}
            for name, desc, sim in specs:
                if not want_leg(name):
                    continue
                try:
                    lp = max(P // 4, 1 << 16)
                    lw = Workload(text, lp, seed=2000 + len(legs), **sim)
                    lpar = H.AlignParams(lw.max_len, 750, 130, 130, H.MmpParams.nt2())
                    lctx = H.Context(gix)
                    step(lctx, lw, lpar)
                    el, la = run_steps(lctx, lw, lpar, 2)
                    legs[name] = {"workload": desc, "pairs_per_step": lp, "pairs_per_s": 2 * lp / el, "ms_per_step": el / 2 * 1e3,
                                  "kernel_ms_per_step": {k[3:]: la[k] / 2 for k in la if k.startswith("ms_")},
                                  "per_pair": {"candidates": la["cand"] / (2 * lp), "pair_hits": la["pair_hits"] / (2 * lp), "failed_pairs": la["failed"] / (2 * lp),
                                               "single_dp_tasks": la["stasks"] / (2 * lp), "rank_blocks_loaded": la["n_occ_loaded"] / (2 * lp)}}
                    lctx.close()
                    del lw
                except Exception as e:  # noqa: BLE001
                    legs[name] = {"workload": desc, "error": repr(e)}
                    log("[bench] leg %s failed: %r" % (name, e))
            # ---- lean layout: what a chunk of a 250 GB NT replica gets -- rank blocks, 13-mer table, the file's SA samples, text
            # (1.09 B/base), none of the optional tables -- and N such chunks resident at once, every batch aligned against each
            try:
                if not want_leg("lean_layout"):
                    raise KeyError("skipped")
                knobs = {"MP_KMER_FILTER": "0", "MP_KMER15": "0", "MP_LKT16": "0", "MP_SA_DENSE": "0"}
                saved = {k: os.environ.get(k) for k in knobs}
                os.environ.update(knobs)
                lean = [H.Index.from_arrays(local_rank, **arrs)]
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                lean_bytes = sum(int(x.info.device_bytes) for x in lean)
                lp = max(P // 4, 1 << 16)
                lw = WorkloadSlice(nt_wl, 0, min(lp, nt_wl.n_pairs))
                params_default, params = params, nt_params                       # (the chunk loop below uses `params`)
                lctx = H.Context(lean[0])
                step(lctx, lw, params)
                el, la = run_steps(lctx, lw, params, 2)
                legs["lean_layout"] = {"workload": "the default workload against ONE index in the lean layout (no 17-mer pre-filter, no 16-mer table, SA samples every 16th row as in "
                                                   "the index files): %.2f B/base in HBM" % (lean[0].info.device_bytes / n_bases),
                                       "index_bytes_in_hbm": int(lean[0].info.device_bytes), "pairs_per_step": lp, "pairs_per_s": 2 * lp / el, "ms_per_step": el / 2 * 1e3,
                                       "kernel_ms_per_step": {k[3:]: la[k] / 2 for k in la if k.startswith("ms_")},
                                       "per_pair": {"rank_blocks_loaded": la["n_occ_loaded"] / (2 * lp), "lkt13_lookups": la["n_lkt"] / (2 * lp), "sa_reads": la["n_sa"] / (2 * lp)}}

                lctx.close()
                for x in lean:
                    x.close()
            except KeyError:
                pass
            except Exception as e:  # noqa: BLE001
                legs["lean_layout"] = {"error": repr(e)}
                log("[bench] lean-layout leg failed: %r" % (e,))
            # ---- the metric's own configuration: a full-size NT replica resident in this GPU's HBM (BASELINE configs[2])
            if want_leg("nt_replica") and args.nt_replica_gb > 0:
                try:
                    ctx.close(); gix.close()                       # the 111 GB every-table layout of the human index makes room
                    ctx = gix = None
                    legs["nt_replica"] = nt_replica_leg(args, H, index_builder, simulate, arrs, text, dev, local_rank, n_bases)
                except Exception as e:  # noqa: BLE001
                    legs["nt_replica"] = {"error": repr(e)}
                    log("[bench] NT replica leg failed: %r" % (e,))
            if "e2e_cli" in out:
                legs["e2e_cli"] = out.pop("e2e_cli")
            out["legs"] = legs
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if gix is not None:
        gix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
