#!/usr/bin/env python3
"""bench.py -- paired 150 bp reads aligned per second against the NT 2BWT index on MI355X (BASELINE.json metric).

Headline (N = 1 and every rank of N > 1): BASELINE configs[2] per GPU -- an NT-sized replica of the database (~89 lean index chunks of
2.95 Gbp, ~243 GB) resident in the GPU's HBM, every batch of synthetic 2x150 bp pairs aligned against EVERY chunk in turn, exactly what
`bin/soap4-mi355x pair <chunk 0>,...,<chunk n-1> ... -b -F -C soap4-nt2.ini -top 95` does for the NT stage of runMegaPath.sh:171-227.

One process per GPU (`torchrun`-style env: RANK / LOCAL_RANK / WORLD_SIZE; `--gpus N` without a launcher starts the N ranks itself).
Reads shard across ranks, the index is replicated into every GPU's HBM, no data-path collective: `"scaling":
"weak"`; the per-reference entry counts of a step are what the ranks all-reduce afterwards (RCCL), as the `reassign` pass 1 they feed would.

A *step* is one batch through all chunks: upload of the packed reads (pinned host memory -> HBM); per chunk the deep-DP round (MMP seeding,
locate, merge, pairing, both DP legs with traceback, hit assembly), the rescue round (single-end DP, mate rescue) and the reduction of the
chunk's hits to per-reference scores on the device; the last chunk's three hit lists to pinned host memory (the BAM files that remain after
the chain are the last chunk's); one sorted key list per batch to pinned host memory.

Prints ONE JSON line on rank 0.  Besides the contract fields:
  roofline        lean seed stage (seed_kernel + locate_kernel per chunk), priced with the bytes of the events that really load memory;
                  `traffic` = HBM bytes per launch from rocprofv3 counters collected LIVE in two child runs of this script (FETCH_SIZE,
                  WRITE_SIZE in separate passes)
  cpu_baseline    the real reference soap4 (oracle/_ref, when it travelled with the snapshot) on the host cores: one chunk's rate / chunks
  parity_sample   the real soap4 chained over two 2.95 Gbp chunks against the product executable with both resident, record by record
  legs            human_filter = BASELINE configs[1] (ONE 2.95 Gbp index, every optional table, soap4.ini: the headline of rounds 1-3) with
                  its own roofline, CPU baseline and parity sample; the C3 / C4 / C5 read shapes; e2e_cli = through the executable
The process exits with code 3 when a parity sample differs (the JSON line is still printed, with "parity_ok": false).
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
def pmc_pass(counter, child_args, tmpdir):
    """Runs this script once more under rocprofv3 with ONE counter (separate passes, MI355X_MICROARCH.md; the program itself follows
    `--`) and returns {kernel: [value per launch]}.  The child does one single-context step of the same workload."""
    out_dir = os.path.join(tmpdir, "pmc_" + counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-include-regex", PMC_REGEX, "--output-format", "csv", "-d", out_dir, "--",
           sys.executable, os.path.abspath(__file__)] + list(child_args)
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
                name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()       # "void merge_kernel<true>(...)" -> "merge_kernel"
                per.setdefault(name, []).append(float(row["Counter_Value"]))
    log("[bench] PMC pass %s: %.0f s, kernels %s" % (counter, time.time() - t0, {k: len(v) for k, v in per.items()}))
    return per


def live_traffic(child_args, tmpdir, kernels=SEED_STAGE_KERNELS):
    if shutil.which("rocprofv3") is None:
        return None
    fetch = pmc_pass("FETCH_SIZE", child_args, tmpdir)
    write = pmc_pass("WRITE_SIZE", child_args, tmpdir) if fetch else None
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
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate child runs of this script (one single-context step each), last launch of each "
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
class NtReplica:
    """BASELINE configs[2] on one GPU: an NT-sized replica of the database resident in HBM in the lean layout (0.92 B/base + a 13-mer
    table per chunk), every batch aligned against every chunk in turn (runMegaPath.sh:191-227 in one process).  Chunk 0 is the genome
    the reads come from; every other chunk is a DIFFERENT genome -- one unrelated synthetic genome of the same size, indexed once and
    copied device-to-device into its own arrays (mp_index_clone) -- so that, as with the real database, a read has its true
    alignment in one chunk and only chance matches in the others."""

    KNOBS = {"MP_KMER_FILTER": "0", "MP_KMER15": "0", "MP_LKT16": "0", "MP_SA_DENSE": "0"}      # the lean layout: none of the optional tables
    LKT14 = os.environ.get("MP_BENCH_NT_LKT14", "1")      # ... and the 13-mer table as 8-byte entries with a seed's first extension step in them (+256 MiB per chunk)

    def __init__(self, args, H, index_builder, simulate, dev, device, n_bases, max_chunks=None, keep_arrays=False):
        self.chunks = []
        self.arrs_b = None
        saved = {k: os.environ.get(k) for k in self.KNOBS}
        saved["MP_LKT14"] = os.environ.get("MP_LKT14")
        os.environ.update(dict(self.KNOBS, MP_KMER_FILTER=os.environ.get("MP_BENCH_NT_KMER", "0"), MP_LKT14=self.LKT14))
        try:
            t0 = time.time()
            self.text_a = simulate.genome(n_bases, seed=3, device=dev)
            bounds = np.linspace(0, n_bases, 25).astype(np.int64)
            ta = self.text_a.cpu().numpy()
            self.arrs_a = index_builder.build_index_arrays([("chr%d" % (i + 1), ta[bounds[i]:bounds[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
            self.build_info = dict(index_builder.last_build_info or {})
            del ta
            torch.cuda.empty_cache()
            self.chunks.append(H.Index.from_arrays(device, **self.arrs_a))
            text_b = simulate.genome(n_bases, seed=4, device=dev)
            tb = text_b.cpu().numpy()
            del text_b
            arrs_b = index_builder.build_index_arrays([("ntB%d" % (i + 1), tb[bounds[i]:bounds[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
            del tb
            torch.cuda.empty_cache()
            self.chunks.append(H.Index.from_arrays(device, **arrs_b))
            if keep_arrays:
                self.arrs_b = arrs_b
            del arrs_b
            per_chunk = int(self.chunks[0].info.device_bytes)
            free_b, total_b = torch.cuda.mem_get_info()
            lean = per_chunk - ((1 << 28) + 8 if self.LKT14 != "0" else 0)                     # --nt-replica-gb counts the database in the lean layout proper
            want = int(args.nt_replica_gb * 1e9 // lean) + 1
            fit = 2 + max(0, int((free_b - (int(args.nt_headroom_gb) << 30)) // per_chunk))     # the rest stays free for the batch buffers and traceback slabs
            self.n_chunks = max(2, min(want, fit, max_chunks or (1 << 30)))
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # every rank holds the same database: the smallest replica any rank has room for (the ranks' counters are summed reference by reference afterwards)
                t = torch.tensor([self.n_chunks], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                self.n_chunks = int(t.item())
            t1 = time.time()
            while len(self.chunks) < self.n_chunks:
                self.chunks.append(self.chunks[1].clone())
            torch.cuda.synchronize()
            self.load_s = time.time() - t0
            self.index_bytes = sum(int(x.info.device_bytes) for x in self.chunks)
            self.hbm_total = int(total_b)
            self.n_bases = n_bases
            log("[bench] NT replica: %d chunks of %.0f Mbp, %.1f GB of index in HBM (two genomes built in %.1f s, %d device-to-device copies in %.1f s)"
                % (self.n_chunks, n_bases / 1e6, self.index_bytes / 1e9, t1 - t0, self.n_chunks - 2, time.time() - t1))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def close(self):
        for x in self.chunks:
            try:
                x.close()
            except Exception:  # noqa: BLE001
                pass
        self.chunks = []


NT_KEYS = ("seed", "locate", "merge", "pair", "fill", "trace", "rescue_tasks", "rescue_fill", "rescue_trace")
NT_EVENTS = ("n_occ_loaded", "n_lkt", "n_sa", "n_seedpos", "n_seeds", "n_hits", "n_lf_steps")


TOP_FLOOR = 0 if os.environ.get("MP_TOP_FLOOR") == "0" else 95          # the NT stage runs with -top 95 (runMegaPath.sh:199); MP_TOP_FLOOR=0: one pass, every rescue round in full


def nt_step(ctx, wl, chunks, params, acc=None, bam=True, mode=1, top_floor=TOP_FLOOR):
    """One batch through every resident chunk, as bin/soap4-mi355x runs the NT stage's command line (`-b ... -F -top 95`, runMegaPath.sh:199):
    upload of the packed reads; per chunk the deep-DP round, the rescue round and the reduction of the chunk's hits to per-mate
    (reference, score) keys on the device; the LAST chunk's three hit lists also travel to the host (the BAM side files that remain after
    the chain are the last chunk's); after the last chunk one sort of the batch's keys and their way to pinned host memory.  With more than one chunk the
    executable makes two passes (the -top rule across chunks: the rescue rounds start once the best score of every read over the deep-DP rounds of ALL chunks is
    known, and leave out the pairs whose rescue work cannot reach top% of it) -- same output, see kernels_floor.hip."""
    n_chunks = len(chunks)
    ctx.set_reads(wl.packed_np, wl.lens_np)
    ctx.reduce_begin(n_chunks)
    cls = np.zeros(wl.n_pairs + 1, dtype=np.uint8)

    def note(ci, st, nh, rst, red):
        if acc is None:
            return
        grp = acc["source_chunk" if ci == 0 else "other_chunks"]
        if st is not None:
            for k, v in zip(NT_KEYS[:6], (st.seed.ms_seed, st.seed.ms_locate, st.seed.ms_merge, st.ms_pair, st.ms_dp_fill, st.ms_dp_trace)):
                grp[k] = grp.get(k, 0.0) + float(v)
            for k in NT_EVENTS:
                grp[k] = grp.get(k, 0) + int(getattr(st.seed, k))
            grp["pair_hits"] = grp.get("pair_hits", 0) + int(nh)
            grp["chunks"] = grp.get("chunks", 0) + 1
        if rst is not None:
            for k, v in zip(NT_KEYS[6:], (rst.ms_tasks, rst.ms_dp_fill, rst.ms_dp_trace)):
                grp[k] = grp.get(k, 0.0) + float(v)
            for k, v in (("single_dp_tasks", rst.n_single_tasks), ("single_hits", rst.n_single_hits), ("rescue_hits", rst.n_rescue_hits), ("failed_pairs", rst.n_failed_pairs),
                         ("floor_pairs", rst.n_floor_pairs), ("floor_tasks", rst.n_floor_tasks), ("floor_mate_tried", rst.n_floor_mate_tried), ("floor_mate_pairs", rst.n_floor_mate_pairs)):
                grp[k] = grp.get(k, 0) + int(v)
        grp["entries"] = grp.get("entries", 0) + int(red[3])

    def whole(ci):
        last = ci + 1 == n_chunks
        ctx.set_index(chunks[ci])
        ctx.set_hit_copies(bam and last)
        _, nh, st = ctx.deep_dp_round(params, fetch=False)
        _, _, rst = ctx.rescue_round(params, False, fetch=False)
        red = ctx.reduce_chunk(mode, ci, cls if last else None)
        if bam and last:
            ctx.pair_hits_view(); ctx.single_hits_view(); ctx.rescue_hits_view()
        note(ci, st, nh, rst, red)
    if not top_floor or n_chunks < 2:
        for ci in range(n_chunks):
            whole(ci)
    else:
        # two passes (the -top rule across chunks, kernels_floor.hip): the deep-DP round of every chunk first, the last chunk as ever, then the rescue rounds
        ctx.set_hit_copies(False)
        for ci in range(n_chunks - 1):
            ctx.set_index(chunks[ci])
            _, nh, st = ctx.deep_dp_round(params, fetch=False)
            red = ctx.reduce_chunk(mode, ci, None, parts=1)
            ctx.chunk_stash(ci)
            note(ci, st, nh, None, red)
        whole(n_chunks - 1)
        ctx.set_hit_copies(False)
        for ci in range(n_chunks - 1):
            ctx.set_index(chunks[ci])
            ctx.chunk_unstash(ci)
            ctx.set_score_floor(top_floor)
            _, _, rst = ctx.rescue_round(params, False, fetch=False)
            ctx.set_score_floor(0)
            red = ctx.reduce_chunk(mode, ci, None, parts=2)
            note(ci, None, 0, rst, red)
    return ctx.reduce_finish()


def nt_run_contexts(ctxs, slices, chunks, params, n_steps, barrier=None):
    """n_steps steps; the batch cut into len(ctxs) contiguous blocks, one context (own HIP streams, own host thread) per block, each taking
    its block through every chunk.  Returns the wall time between the barriers."""
    import threading
    errs = []

    def worker(i):
        try:
            for _ in range(n_steps):
                nt_step(ctxs[i], slices[i], chunks, params)
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


def nt_seed_bytes(groups, sum_len, n_chunk_steps):
    """Bytes the lean seed stage (seed_kernel + expand_hits_kernel + locate_kernel) requests from memory, by event, each counted by the kernels
    themselves: 64 B per rank block loaded (the seeding walk's and the LF walk's), 8 B per 13-mer table look-up (two adjacent 32-bit bounds),
    4 B per sampled suffix-array entry, 0.25 B per read base, 16 B per kept seed position written."""
    ev = {k: sum(g.get(k, 0) for g in groups) for k in ("n_occ_loaded", "n_lkt", "n_sa", "n_seedpos")}
    return 64 * ev["n_occ_loaded"] + 8 * ev["n_lkt"] + 4 * ev["n_sa"] + 0.25 * sum_len * n_chunk_steps + 16 * ev["n_seedpos"]


def nt_cpu_baseline(args, index_builder, rep, read_len, n_pairs, lkt_b):
    """The real soap4 beside it: the same read mix against ONE of the other chunks (its files written here), on the host cores; a database of
    n chunks costs it n such runs (runMegaPath.sh:191-227), so its rate for the replica is that of one run / n."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "soap4")
    if not os.path.exists(ref_bin) or rep.arrs_b is None:
        return None, None
    td = tempfile.mkdtemp(prefix="mpnt_")
    try:
        prefix_b = os.path.join(td, "ntB.index")
        index_builder.write_index_files(prefix_b, rep.arrs_b, write_lkt=True, lkt_table=lkt_b)
        cw = Workload(rep.text_a, n_pairs, seed=4242, read_len=read_len, L=read_len + 1, sub_max=0.08, random_frac=0.4)
        f1, f2 = os.path.join(td, "n_1.fq"), os.path.join(td, "n_2.fq")
        write_fastq(cw.codes.cpu().numpy(), n_pairs, f1, f2)
        ini = os.path.join(td, "soap4-nt2.ini")
        with open(ini, "w") as f:
            f.write(NT2_INI)
        T = min(os.cpu_count() or 1, 32)
        with open(os.devnull, "wb") as fo, open(os.path.join(td, "err"), "wb") as fe:
            rc = subprocess.call([ref_bin, "pair", prefix_b, f1, f2, "-b", "-o", os.path.join(td, "o"), "-L", str(read_len + 1), "-T", str(T), "-u", "750", "-F", "-C", ini, "-top", "95", "-nc"],
                                 stdout=fo, stderr=fe)
        al = None
        for line in open(os.path.join(td, "err"), errors="replace"):
            if "Overall alignment time" in line:
                al = float(line.split(":")[1].split()[0])
        if rc != 0 or not al:
            log("[bench] NT cpu baseline: the reference soap4 failed (rc %d)" % rc)
            return None, td
        return {"kind": "reference", "threads": T, "pairs": n_pairs, "alignment_s_one_chunk": al, "pairs_per_s_one_chunk": n_pairs / al}, td
    except Exception as e:  # noqa: BLE001
        log("[bench] NT cpu baseline failed: %r" % (e,))
        return None, td


def deinterleave(src, out1, out2):
    lines = open(src, "rb").read().split(b"\n")
    with open(out1, "wb") as a, open(out2, "wb") as b:
        for i in range(0, len(lines) - 7, 8):
            a.write(b"\n".join(lines[i:i + 4]) + b"\n")
            b.write(b"\n".join(lines[i + 4:i + 8]) + b"\n")


def nt_parity_sample(args, index_builder, rep, td, device, read_len, n_pairs, lkt_a):
    """Parity at chunk size: the REAL soap4 chained over two 2.95 Gbp chunks -- chunk 0 (the reads' genome) and one foreign chunk -- with the
    NT stage's own command lines (runMegaPath.sh:199-208: -b -o <same prefix> ... -F -C soap4-nt2.ini -top 95, -nc for the first run only,
    the second run reads the de-interleaved stdout of the first), against bin/soap4-mi355x with the two chunks resident in one process."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "soap4")
    exe = os.path.join(ROOT, "bin", "soap4-mi355x")
    if td is None or not os.path.exists(ref_bin) or not os.path.exists(exe):
        return None
    t_start = time.time()
    prefix_a, prefix_b = os.path.join(td, "ntA.index"), os.path.join(td, "ntB.index")
    index_builder.write_index_files(prefix_a, rep.arrs_a, write_lkt=True, lkt_table=lkt_a)
    cw = Workload(rep.text_a, n_pairs, seed=5151, read_len=read_len, L=read_len + 1, sub_max=0.08, random_frac=0.4)
    f1, f2 = os.path.join(td, "S.nt.tmp_in_1.fq"), os.path.join(td, "S.nt.tmp_in_2.fq")
    write_fastq(cw.codes.cpu().numpy(), n_pairs, f1, f2)
    ini = os.path.join(td, "soap4-nt2.ini")
    T = min(os.cpu_count() or 1, 32)
    common = ["-b", "-o", "S.nt.bam", "-L", str(read_len + 1), "-T", str(T), "-u", "750", "-F", "-C", ini, "-top", "95"]
    # the product first (it reads the original files; the chain below moves them)
    prod_out = os.path.join(td, "prod.out")
    env = dict(os.environ, MP_DEVICE=str(device), MP_LKT14=NtReplica.LKT14, **NtReplica.KNOBS)
    env.pop("LOCAL_RANK", None)
    with open(prod_out, "wb") as fo, open(os.path.join(td, "prod.err"), "wb") as fe:
        rc = subprocess.call([exe, "pair", prefix_a + "," + prefix_b, "S.nt.tmp_in_1.fq", "S.nt.tmp_in_2.fq"] + common + ["-nc"], stdout=fo, stderr=fe, env=env, cwd=td)
    if rc != 0:
        return {"pairs": n_pairs, "records_equal": False, "first_diff": "product executable failed (rc %d): %s" % (rc, open(os.path.join(td, "prod.err"), errors="replace").read()[-300:])}
    t_prod = time.time() - t_start
    ref_out = None
    for i, prefix in enumerate((prefix_a, prefix_b)):
        ref_out = os.path.join(td, "ref_%d.out" % i)
        with open(ref_out, "wb") as fo, open(os.path.join(td, "ref.err"), "wb") as fe:
            rc = subprocess.call([ref_bin, "pair", prefix, "S.nt.tmp_in_1.fq", "S.nt.tmp_in_2.fq"] + common + (["-nc"] if i == 0 else []), stdout=fo, stderr=fe, cwd=td)
        if rc != 0:
            return {"pairs": n_pairs, "records_equal": None, "first_diff": "reference soap4 failed on chunk %d (rc %d)" % (i, rc)}
        deinterleave(ref_out, f1, f2)
    want, got = fastq_records(ref_out), fastq_records(prod_out)
    diff = [k for k in want if got.get(k) != want[k]] + [k for k in got if k not in want]
    first = None
    if diff:
        k = sorted(diff)[0]
        first = {"record": "%s #%d" % (k[0].decode(), k[1]), "reference": want.get(k, (b"<absent>",))[0].decode(errors="replace")[:300],
                 "product": got.get(k, (b"<absent>",))[0].decode(errors="replace")[:300]}
    two = sum(1 for v in want.values() if v[0].count(b",chr") + v[0].count(b",ntB") >= 2)
    return {"pairs": n_pairs, "records": len(want), "records_equal": not diff, "records_differing": len(diff), "first_diff": first,
            "reference_records_with_a_hit": sum(1 for v in want.values() if b"SCORE:0;" not in v[0]), "reference_records_listing_two_or_more_references": two, "against": "reference",
            "how": "the real soap4 chained over chunk 0 and one foreign chunk of %.0f Mbp each (runMegaPath.sh:199-208 command lines: -b -o <same prefix> -L %d -T %d -u 750 -F -C soap4-nt2.ini "
                   "-top 95, -nc for the first run, the second run reads the de-interleaved stdout of the first) against bin/soap4-mi355x pair <chunk0>,<chunk1> with the same options "
                   "(both chunks resident, %.0f s whole-process); last stdout compared record by record, order-insensitive; whole sample %.0f s"
                   % (rep.n_bases / 1e6, read_len + 1, T, t_prod, time.time() - t_start)}


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

        def run(a, b, n_pairs, extra_env=None, threads=threads):
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
        # twice: the first process after the files were written runs on cold pages, and on a box whose CPU share is a cgroup quota (16 of 256 cores here) a run can
        # lose seconds to throttling; the faster run is quoted, the other one listed
        runs = [run(f1, f2, n_batches * P), run(f1, f2, n_batches * P)]
        runs.sort(key=lambda r: -(r.get("pairs_per_s") or 0))
        res["plain"], res["plain_other_run"] = runs[0], runs[1]
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
                for label, extra, flags in (("device_reduce", {}, []), ("host_merge", {"MP_HOST_MERGE": "1"}, []),
                                            ("bam_device_reduce", {}, ["-b"]), ("bam_host_merge", {"MP_BAM_HOST_MERGE": "1"}, ["-b"])):
                    cmd = [exe, "pair", ",".join([prefix] * k), n1, n2, "-o", os.path.join(d, "o"), "-C", nt_ini, "-L", str(args.read_len + 1), "-T", str(threads), "-u", "750", "-F", "-nc", "-top", "95"] + flags
                    with open(os.devnull, "wb") as fo:
                        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=dict(env, **extra))
                    err = r.stderr.decode(errors="replace")
                    loop_s = next((float(l.split("alignment loop")[1].split()[0]) for l in err.splitlines() if "alignment loop" in l), None)
                    rc8["%d_chunks_%s" % (k, label)] = {"rc": r.returncode, "alignment_loop_s": loop_s, "pairs_per_s": ntp / loop_s if loop_s else None,
                                                        "stage_seconds": next((l.split("seconds:")[1].strip() for l in err.splitlines() if "seconds:" in l), None)}
            rc8["input"] = ("%d pairs of the NT read mix (60 %% at 0-8 %% divergence, 40 %% random) against the same lean index given 2 and 8 times; device_reduce = the "
                            "per-reference scores reduced on the device over all chunks, ONE printer pass per batch (mp_reduce_*); host_merge = hit lists to the host, header "
                            "comments merged chunk after chunk (MP_HOST_MERGE=1); bam_* = the same with -b, the NT stage's own flag (runMegaPath.sh:199): by default only the "
                            "LAST chunk's hit lists travel to the host for the BAM records, MP_BAM_HOST_MERGE=1 is the round-3 path (every chunk's).  'output of ...' in "
                            "stage_seconds is the printer's time" % ntp)
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
        # the same at more host threads (-T sets the reader's, the formatters' and the inflaters' share): where does the host side saturate?
        # (a GPU box's share of its host's cores may be smaller than what os.cpu_count() reports: the CPUs this process may run on are listed)
        try:
            usable = len(os.sched_getaffinity(0))
        except Exception:  # noqa: BLE001
            usable = os.cpu_count() or 1
        res["host_cpus"] = {"os_cpu_count": os.cpu_count(), "sched_affinity": usable, "cgroup_cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)}
        for T in (32,):
            if (os.cpu_count() or 1) >= T and T > threads:
                res["gz_T%d" % T] = run(g1 + ".gz", g2 + ".gz", gz_pairs, {"MP_INFLATE_THREADS": str(min(16, T * 3 // 8))}, threads=T)
                res["plain_T%d" % T] = run(f1, f2, n_batches * P, threads=T)
        if os.environ.get("MP_BENCH_E2E_FULL") == "1":                                            # the round-3 comparisons, not part of the default run
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


# ------------------------------------------------------------------------------------------------ configs[1] and the read-shape legs
def human_filter_leg(args, H, index_builder, simulate, text, arrs, dev, device, rank, n_bases, want_leg, child_args_base):
    """BASELINE configs[1] as the pipeline runs it (runMegaPath.sh:136: -C soap4.ini -L <read length>): the every-table layout of ONE 2.95 Gbp
    index (111 GB), 1 Mi pairs per step.  Until round 3 this was the bench's headline; it is now a leg with its own small roofline, its CPU
    baseline and its parity sample.  Returns (leg dict, sub-legs dict, parity ok)."""
    legs = {}
    torch.cuda.empty_cache()
    t1 = time.time()
    gix = H.Index.from_arrays(device, **arrs)
    log("[bench] human-filter leg: every-table layout of the %.0f Mbp index in HBM in %.1f s, %.2f GB" % (args.genome_mbp, time.time() - t1, gix.info.device_bytes / 1e9))
    torch.cuda.empty_cache()
    g_ms, g_bytes = gix.bench_gather64()
    gather_gbs = g_bytes / (g_ms * 1e-3) / 1e9
    P = args.pairs_per_step
    K = max(1, args.leg_steps)
    read_L = args.L if args.L is not None else (args.read_len if args.ini == "soap4.ini" else args.read_len + 1)
    wl = Workload(text, P, seed=1000 + rank, read_len=args.read_len, L=read_L)
    ctx = H.Context(gix)
    params = H.AlignParams(wl.max_len, 750, 130, 130, mmp_of(args.ini))       # insert_low = the longest read of the first batch (SOAP4.cpp:465-474)
    step(ctx, wl, params)
    ctx.set_dp_census(True)
    step(ctx, wl, params)
    census = ctx.dp_census()
    ctx.set_dp_census(False)
    band0, th0 = ctx.dp_band_stats(), ctx.dp_thermo_stats()
    elapsed_single, acc = run_steps(ctx, wl, params, K)
    band1, th1 = ctx.dp_band_stats(), ctx.dp_thermo_stats()
    elapsed = elapsed_single
    n_ctx = max(1, args.leg_contexts)
    if n_ctx > 1:
        cuts = [P * i // n_ctx for i in range(n_ctx + 1)]
        slices = [WorkloadSlice(wl, cuts[i], cuts[i + 1]) for i in range(n_ctx)]
        ctxs = [ctx] + [H.Context(gix) for _ in range(n_ctx - 1)]
        run_overlapped(ctxs, slices, params, 1)                      # first touch of the new contexts' buffers
        elapsed = run_overlapped(ctxs, slices, params, K)
        for c in ctxs[1:]:
            c.close()
    t_seed_s = (acc["ms_seed"] + acc["ms_locate"]) / 1e3
    fused = acc["ms_locate"] / K < 0.05          # whole suffix array resident: no expand / locate launch (merge_kernel<true> makes its keys from the seed records)
    roof_kernels = ("seed_kernel", "seed_overflow_kernel") if fused else SEED_STAGE_KERNELS
    b_loaded = loaded_bytes(acc, wl.sum_len, K, fused)
    achieved = b_loaded / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
    per_pair = lambda k: acc[k] / (K * P)      # noqa: E731
    leg = {
        "workload": "2x%dbp synthetic pairs vs ONE synthetic %.0f Mbp genome in 24 sequences, every optional table resident (BASELINE configs[1], human-filter stage; GRCh38 indexes "
                    "~2.95 Gbp of non-N bases), %d pairs/step; a step = upload of the packed reads + deep-DP round + rescue round + all three hit lists back in pinned host memory; "
                    "%d contexts side by side" % (args.read_len, args.genome_mbp, P, n_ctx),
        "command_line": "soap4 pair <index> r1.fq r2.fq -C %s -L %d -u 750 -F -nc (runMegaPath.sh:136)" % (args.ini, read_L),
        "pairs_per_s": K * P / elapsed, "ms_per_step": elapsed / K * 1e3, "steps": K, "contexts": n_ctx, "index_bytes_in_hbm": int(gix.info.device_bytes),
        "single_context": {"pairs_per_s": K * P / elapsed_single, "ms_per_step": elapsed_single / K * 1e3},
        "kernel_ms_per_step": {k[3:]: acc[k] / K for k in acc if k.startswith("ms_")},
        "roofline": {"bound": "hbm", "kernel": "+".join(roof_kernels), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "bytes_per_launch": b_loaded / K, "bytes_per_pair": b_loaded / (K * P), "launch_ms": (acc["ms_seed"] + acc["ms_locate"]) / K,
                     "random64_gather_ceiling_gbs": gather_gbs,
                     "formula": "64*rank_blocks_loaded + 8*(lkt13 + lkt16 look-ups) + 8*prefilter_entries_loaded + 4*sa_reads + 8*text_words_compared + 0.25*read_bases + 16*seed_positions_out"},
        "events_per_pair": {"rank_blocks_loaded": per_pair("n_occ_loaded"), "lkt13_lookups": per_pair("n_lkt"), "lkt16_lookups": per_pair("n_lkt16"), "sa_reads": per_pair("n_sa"),
                            "prefilter_entries_loaded": per_pair("n_prefilter_loads"), "text_words_compared": per_pair("n_text_words"), "seeds": per_pair("n_seeds"),
                            "seed_positions_out": per_pair("n_seedpos"), "candidates": per_pair("cand"), "pair_hits": per_pair("pair_hits"), "failed_pairs": per_pair("failed"),
                            "single_dp_tasks": per_pair("stasks")},
    }
    t_dp_s = (acc["ms_fill"] + acc["ms_rescue_fill"]) / 1e3 / K
    if t_dp_s > 0:
        computed = census.cells_band_score + census.cells_band_trace + census.cells_full_score + census.cells_full_trace
        leg["dp"] = {"bound": "valu", "unit": "Gcells/s", "achieved": computed / t_dp_s / 1e9, "instances_per_step": int(census.instances),
                     "cells_computed_per_step": {"band_score_pass": int(census.cells_band_score), "band_traceback_pass": int(census.cells_band_trace),
                                                 "full_table_score_only": int(census.cells_full_score), "full_table_with_traceback": int(census.cells_full_trace)},
                     "full_table_equivalent_gcells_per_s": census.cells_full_table / t_dp_s / 1e9,
                     "band_instances_per_step": (band1[0] - band0[0]) / K, "band_left_to_full_table_per_step": (band1[1] - band0[1]) / K,
                     "rescue_certificate_per_step": dict(zip(("certified_no_alignment", "score_of_10_elsewhere", "bound_reaches_cutoff", "not_eligible"), [(b - a) / K for a, b in zip(th0, th1)])),
                     "fill_ms_per_step": t_dp_s * 1e3}
    parity_ok = True
    tmpdir = tempfile.mkdtemp(prefix="mpbench_")
    try:
        if not args.no_pmc and want_leg("human_filter_pmc"):
            try:
                tr = live_traffic(child_args_base + ["--pmc-child", "human"], tmpdir, roof_kernels)
            except Exception as e:  # noqa: BLE001
                log("[bench] PMC passes of the human-filter leg failed: %r" % (e,))
                tr = None
            if tr:
                rf = leg["roofline"]
                rf["traffic"] = tr["bytes_per_launch"]
                rf["traffic_gbs"] = tr["bytes_per_launch"] / (rf["launch_ms"] / 1e3) / 1e9
                rf["frac_of_gather_ceiling"] = rf["traffic_gbs"] / gather_gbs
                rf["traffic_over_requested"] = tr["bytes_per_launch"] / rf["bytes_per_launch"]
        if not args.no_cpu_baseline:
            try:
                cb, parity = cpu_baseline_and_parity(args, arrs, wl.codes[: 2 * min(args.cpu_sample_pairs, P)], tmpdir, gix.lkt_table(), device, read_L,
                                                     threads_list=[min(os.cpu_count() or 1, 32)])
                leg["cpu_baseline"] = cb
                leg["parity_sample"] = parity
                if parity and parity.get("records_equal") is False:
                    parity_ok = False
            except Exception as e:  # noqa: BLE001 -- auxiliary; report its absence
                log("[bench] cpu baseline / parity sample of the human-filter leg failed: %r" % (e,))
        if not args.no_e2e and want_leg("e2e_cli"):
            try:
                prefix = os.path.join(tmpdir, "ref.index")
                if not os.path.exists(prefix + ".bwt"):
                    index_builder.write_index_files(prefix, arrs, write_lkt=False)
                ini_path = os.path.join(tmpdir, args.ini)
                with open(ini_path, "w") as f:
                    f.write(INIS[args.ini])
                legs["e2e_cli"] = e2e_cli_leg(args, text, prefix, ini_path, read_L, device, tmpdir, n_batches=args.e2e_batches)
            except Exception as e:  # noqa: BLE001
                legs["e2e_cli"] = {"error": repr(e)}
                log("[bench] e2e leg failed: %r" % (e,))
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    # ---- the other configs' read shapes against the same index, NT-stage parameters (soap4-nt2.ini, -L = read length + 1, runMegaPath.sh:206)
    specs = [
        ("C3_divergent", "2x150 bp, per-pair divergence uniform 0-8 %, 40 % random pairs (NT-stage read mix)", dict(read_len=150, sub_max=0.08, random_frac=0.4)),
        ("C4_2x250", "2x250 bp, -L 251, fragments ~N(450,50): 311-row windows, 32-lane DP shapes", dict(read_len=250, frag_mean=450.0, frag_sd=50.0, frag_max=700)),
        ("C5_mixed_rescue", "lengths uniform 75-300 (-L 301), 30 % of pairs with one mate random: rescue stress", dict(
            read_len=300, len_range=(75, 300), broken_mate_frac=0.3, frag_mean=450.0, frag_sd=50.0, frag_max=700)),
    ]
    nt_L = args.read_len + 1
    nt_wl = Workload(text, max(P // 4, 1 << 16), seed=1000 + rank, read_len=args.read_len, L=nt_L)
    nt_params = H.AlignParams(nt_wl.max_len, 750, 130, 130, H.MmpParams.nt2())
    if want_leg("nt2_ini_same_reads"):
        try:
            lp = nt_wl.n_pairs
            lctx = H.Context(gix)
            step(lctx, nt_wl, nt_params)
            el, la = run_steps(lctx, nt_wl, nt_params, 2)
            legs["nt2_ini_same_reads"] = {"workload": "the human-filter leg's read mix with the NT stage's parameters (-C soap4-nt2.ini -L %d: 17-base seeds) against the same index, one context" % nt_L,
                                          "pairs_per_step": lp, "pairs_per_s": 2 * lp / el, "ms_per_step": el / 2 * 1e3, "kernel_ms_per_step": {k[3:]: la[k] / 2 for k in la if k.startswith("ms_")}}
            lctx.close()
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
    ctx.close()
    gix.close()
    return leg, legs, parity_ok


# ------------------------------------------------------------------------------------------------ main
def main():
    if sys.argv[1:2] == ["--write-bgzf"]:                       # helper mode of the e2e leg (no GPU is touched)
        write_bgzf(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("MP_BENCH_GENOME_MBP", "2950")), help="bases per index chunk (< 4295 Mbp: 32-bit positions per chunk)")
    ap.add_argument("--nt-replica-gb", type=float, default=float(os.environ.get("MP_BENCH_NT_GB", "242")),
                    help="HBM bytes (GB) of index the replica makes resident (as many lean chunks of --genome-mbp as that takes and as fit)")
    ap.add_argument("--nt-headroom-gb", type=float, default=float(os.environ.get("MP_BENCH_NT_HEADROOM_GB", "32" if NtReplica.LKT14 != "0" else "56")),
                    help="GiB of HBM kept free of index for the batch buffers (32: room for 90 chunks with the packed 14-mer tables; the 89 of the default replica are 266.6 GB and leave 9 GB free during the steps)")
    ap.add_argument("--nt-pairs", type=int, default=int(os.environ.get("MP_BENCH_NT_PAIRS", str(1 << 18))), help="pairs per step and GPU of the headline workload")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("MP_BENCH_CONTEXTS", "4")),
                    help="contexts (HIP stream pairs + host threads) a step's batch is cut across; 1 = one context does the whole batch")
    ap.add_argument("--leg-contexts", type=int, default=int(os.environ.get("MP_BENCH_LEG_CONTEXTS", "3")), help="contexts of the human-filter leg (one index: three, as the executable's workers per GPU)")
    ap.add_argument("--single-steps", type=int, default=int(os.environ.get("MP_BENCH_SINGLE_STEPS", "1")), help="untimed one-context steps the per-kernel HIP-event times and the roofline come from")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--pairs-per-step", type=int, default=int(os.environ.get("MP_BENCH_PAIRS", str(1 << 20))), help="pairs per step of the human-filter leg")
    ap.add_argument("--leg-steps", type=int, default=int(os.environ.get("MP_BENCH_LEG_STEPS", "5")), help="timed steps of the human-filter leg")
    ap.add_argument("--ini", choices=sorted(INIS), default=os.environ.get("MP_BENCH_INI", "soap4.ini"), help="ini of the human-filter leg (soap4.ini + -L <read-len>: runMegaPath.sh:136)")
    ap.add_argument("--L", type=int, default=None, help="-L of the human-filter leg (default: read-len for soap4.ini, read-len + 1 for soap4-nt2.ini)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=int(os.environ.get("MP_BENCH_CPU_PAIRS", "100000")), help="pairs of the human-filter leg's CPU baseline / parity sample")
    ap.add_argument("--nt-cpu-pairs", type=int, default=int(os.environ.get("MP_BENCH_NT_CPU_PAIRS", "20000")), help="pairs of the headline's CPU baseline (one chunk) and of its two-chunk parity sample")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baselines and the parity samples")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 child runs (roofline.traffic = null)")
    ap.add_argument("--no-legs", action="store_true", help="skip every leg (configs[1], read shapes, the executable)")
    ap.add_argument("--legs", default=os.environ.get("MP_BENCH_LEGS", "all"), help="comma-separated legs to run (human_filter, human_filter_pmc, e2e_cli, nt2_ini_same_reads, C3_divergent, "
                    "C4_2x250, C5_mixed_rescue) or 'all'")
    ap.add_argument("--no-e2e", action="store_true", help="skip the leg through the executable (FASTQ files in, annotated FASTQ out)")
    ap.add_argument("--e2e-batches", type=int, default=int(os.environ.get("MP_BENCH_E2E_BATCHES", "12")), help="batches of --pairs-per-step pairs in the e2e leg's input files")
    ap.add_argument("--pmc-child", default=None, choices=["nt", "human"], help=argparse.SUPPRESS)
    args = ap.parse_args()
    leg_set = None if args.legs == "all" else set(x for x in args.legs.split(",") if x)

    def want_leg(name):
        return leg_set is None or name in leg_set

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
    child_args_base = ["--genome-mbp", str(args.genome_mbp), "--nt-pairs", str(args.nt_pairs), "--pairs-per-step", str(args.pairs_per_step), "--read-len", str(args.read_len),
                       "--ini", args.ini, "--nt-headroom-gb", str(args.nt_headroom_gb)] + (["--L", str(args.L)] if args.L is not None else [])

    if args.pmc_child == "human":
        # one step of the human-filter leg's batch under the profiler's counters
        text = simulate.genome(n_bases, seed=3, device=dev)
        bounds = np.linspace(0, n_bases, 25).astype(np.int64)
        tc = text.cpu().numpy()
        arrs = index_builder.build_index_arrays([("chr%d" % (i + 1), tc[bounds[i]:bounds[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
        del tc
        torch.cuda.empty_cache()
        gix = H.Index.from_arrays(local_rank, **arrs)
        gix.bench_gather64()
        read_L = args.L if args.L is not None else (args.read_len if args.ini == "soap4.ini" else args.read_len + 1)
        wl = Workload(text, args.pairs_per_step, seed=1000, read_len=args.read_len, L=read_L)
        ctx = H.Context(gix)
        step(ctx, wl, H.AlignParams(wl.max_len, 750, 130, 130, mmp_of(args.ini)))
        ctx.close(); gix.close()
        return

    # ---------------------------------------------------------------- the headline: the NT replica (BASELINE configs[2] per GPU)
    pmc_nt = args.pmc_child == "nt"
    rep = NtReplica(args, H, index_builder, simulate, dev, local_rank, n_bases, max_chunks=3 if pmc_nt else None,
                    keep_arrays=(rank == 0 and world == 1 and not args.no_cpu_baseline and not pmc_nt))
    if pmc_nt:
        rep.chunks[1].bench_gather64()                                # the calibration launches the traffic figure is read against
    lp = args.nt_pairs
    wl = Workload(rep.text_a, lp, seed=4242 + rank, read_len=args.read_len, L=args.read_len + 1, sub_max=0.08, random_frac=0.4)      # 60 % from chunk 0 at 0-8 % divergence, 40 % random
    params = H.AlignParams(wl.max_len, 750, 130, 130, H.MmpParams.nt2())
    n_chunks = rep.n_chunks
    ctx0 = H.Context(rep.chunks[0])
    if pmc_nt:
        nt_step(ctx0, wl, rep.chunks, params)
        ctx0.close(); rep.close()
        return
    g_ms, g_bytes = rep.chunks[1].bench_gather64()
    gather_gbs = g_bytes / (g_ms * 1e-3) / 1e9
    if rank == 0:
        log("[bench] random 64-B gather ceiling over a chunk's rank blocks: %.0f GB/s" % gather_gbs)
    n_ctx = max(1, args.contexts)
    cuts = [lp * i // n_ctx for i in range(n_ctx + 1)]
    slices = [WorkloadSlice(wl, cuts[i], cuts[i + 1]) for i in range(n_ctx)]
    ctxs = [ctx0] + [H.Context(rep.chunks[0]) for _ in range(n_ctx - 1)]
    K, Wm = args.steps, max(args.warmup, 1)                          # (at least one untimed step: the first touch of every context's buffers)
    t_w = time.time()
    nt_run_contexts(ctxs, slices, rep.chunks, params, Wm)
    if rank == 0:
        log("[bench] %d warm-up step(s) of %d pairs through %d chunks: %.1f s" % (Wm, lp, n_chunks, time.time() - t_w))
    elapsed = nt_run_contexts(ctxs, slices, rep.chunks, params, K, barrier)          # THE timed region: K steps between barriers + device syncs
    hbm_free_timed = int(torch.cuda.mem_get_info()[0])                               # (the contexts' buffers only ever grow: what is free now was free throughout)
    for c in ctxs[1:]:
        c.close()
    # the run the per-kernel HIP-event times, the event counts and the roofline come from: ONE context doing the whole batch on one stream (kernels of
    # several contexts overlap and stretch each other); this is also the last step whose keys feed the exchange step below
    S = max(1, args.single_steps)
    acc = {"source_chunk": {}, "other_chunks": {}}
    th_a = ctx0.dp_thermo_stats()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(S):
        keys = nt_step(ctx0, wl, rep.chunks, params, acc)
    torch.cuda.synchronize()
    el_single = (time.perf_counter() - t3) / S
    th_b = ctx0.dp_thermo_stats()
    # the exchange step of the path: per-reference entry counts of this rank's last step, summed over the ranks (RCCL)
    kk = np.array(keys, dtype=np.uint64)
    n_seq = int(rep.chunks[0].info.num_seq)
    chunk_of = (n_chunks - 1) - ((kk >> np.uint64(32)) & np.uint64(1023)).astype(np.int64)
    seq_of = ((kk >> np.uint64(10)) & np.uint64(0x3FFFFF)).astype(np.int64)
    ref_counts = np.bincount(chunk_of * (n_seq + 1) + seq_of, minlength=n_chunks * (n_seq + 1)).astype(np.int64)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ref_counts = taxcount.allreduce_counts(ref_counts, dev if backend == "nccl" else "cpu")
    ctx0.close()

    parity_ok = True
    if rank == 0:
        o, src = acc["other_chunks"], acc["source_chunk"]
        n_o = max(1, o.get("chunks", 1))
        t_seed_s = (o.get("seed", 0.0) + o.get("locate", 0.0) + src.get("seed", 0.0) + src.get("locate", 0.0)) / 1e3
        n_launch = S * n_chunks
        seed_bytes = nt_seed_bytes([o, src], wl.sum_len, n_launch)
        achieved = seed_bytes / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
        seed_bytes_other = nt_seed_bytes([o], wl.sum_len, n_o)
        out = {
            "metric": "paired 150bp reads aligned/sec vs NT 2BWT index",
            "value": world * K * lp / elapsed,
            "unit": "pairs/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] per GPU: %d index chunks of %.0f Mbp resident at once in the lean layout = %.1f Gbp, %.1f GB of HBM; chunk 0 is the genome the reads "
                                   "come from, the other %d hold an unrelated synthetic genome (indexed once, copied device-to-device into separate arrays), so a read finds its true "
                                   "alignment in one chunk and chance matches elsewhere, as in the real database; 2x%d bp, 60 %% of the pairs from chunk 0 at 0-8 %% divergence, 40 %% "
                                   "random (SURVEY 8d C3); %d pairs/step/GPU; a step = one batch through EVERY chunk as bin/soap4-mi355x runs the NT stage: upload of the packed reads, per "
                                   "chunk deep-DP round + rescue round + reduction of the hits to per-reference scores on the device, the last chunk's three hit lists to the host "
                                   "(the BAM files that remain are the last chunk's), one sorted key list per batch in pinned host memory; the batch is cut into %d contiguous blocks "
                                   "that as many contexts work on side by side" % (n_chunks, n_bases / 1e6, n_chunks * n_bases / 1e9, rep.index_bytes / 1e9, n_chunks - 1,
                                                                                    args.read_len, lp, n_ctx),
                       "command_line": "soap4 pair <chunk i> in_1.fq in_2.fq -b -o <prefix>.nt.bam -L %d -T <n> -u 750 -F -C soap4-nt2.ini -top 95 [-nc for i = 0], i = 0 .. %d "
                                       "(runMegaPath.sh:191-208), as ONE process: bin/soap4-mi355x pair <chunk 0>,...,<chunk %d> ..." % (args.read_len + 1, n_chunks - 1, n_chunks - 1),
                       "index_replicated_per_gpu": True, "chunks": n_chunks, "bases_resident": n_chunks * n_bases, "index_bytes_in_hbm": rep.index_bytes,
                       "bytes_per_base": rep.index_bytes / (n_chunks * n_bases), "hbm_total_bytes": rep.hbm_total, "pairs_per_step_per_gpu": lp, "contexts_per_gpu": n_ctx, "hbm_free_bytes_during_timed_steps": hbm_free_timed,
                       "ini": "soap4-nt2.ini", "L": args.read_len + 1, "u": 750, "top": 95, "replica_load_s": rep.load_s},
            "chunk_pair_alignments_per_s": world * K * lp * n_chunks / elapsed,
            "single_context": {"ms_per_step": el_single * 1e3, "pairs_per_s": lp / el_single, "steps": S,
                               "note": "ONE context taking the whole batch through every chunk on one stream, after the timed region: the run the per-kernel HIP-event times, "
                                       "the event counts and the roofline come from (kernels of several contexts overlap and stretch each other)"},
            "roofline": {"bound": "hbm", "kernel": "seed_kernel+seed_overflow_kernel+expand_hits_kernel+locate_kernel (lean layout: the seed stage K1 + K2 of one chunk)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": seed_bytes / n_launch, "launch_ms": t_seed_s * 1e3 / n_launch, "launches": n_launch,
                         "bytes_per_pair_per_chunk": seed_bytes / (lp * n_launch),
                         "other_chunks": {"bytes_per_launch": seed_bytes_other / n_o, "launch_ms": (o.get("seed", 0.0) + o.get("locate", 0.0)) / n_o},
                         # the two kernels apart (chunks none of the reads comes from): the seeding walk's own blocks and table look-ups over seed_kernel's time, the LF
                         # steps' blocks and the suffix-array reads over expand_hits_kernel + locate_kernel's
                         "seed_kernel_alone": {"gbs": ((64 * (o.get("n_occ_loaded", 0) - o.get("n_lf_steps", 0)) + 8 * o.get("n_lkt", 0)) / (o.get("seed", 0.0) / 1e3) / 1e9) if o.get("seed") else None,
                                               "launch_ms": o.get("seed", 0.0) / n_o},
                         "locate_alone": {"gbs": ((64 * o.get("n_lf_steps", 0) + 4 * o.get("n_sa", 0)) / (o.get("locate", 0.0) / 1e3) / 1e9) if o.get("locate") else None,
                                          "launch_ms": o.get("locate", 0.0) / n_o},
                         "formula": "64*rank_blocks_loaded + 8*lkt13_lookups + 4*sa_reads + 0.25*read_bases + 16*seed_positions_out per chunk-launch of the seed stage (every count emitted by "
                                    "the kernels), over the HIP-event time of those kernels on the context's stream, averaged over the %d chunk-launches of the one-context step(s)" % n_launch,
                         "random64_gather_ceiling_gbs": gather_gbs, "frac_of_gather_ceiling_requested": achieved / gather_gbs,
                         "note": "dependent random 64-byte gathers: the physical ceiling of this access shape is the measured random-64-B-gather rate, not the streaming peak"},
            "kernel_ms_per_chunk_other_chunks": {k: o.get(k, 0.0) / n_o for k in NT_KEYS},
            "kernel_ms_source_chunk": {k: src.get(k, 0.0) / max(1, src.get("chunks", 1)) for k in NT_KEYS},
            "per_pair_per_other_chunk": {k: o.get(k, 0) / (n_o * lp) for k in NT_EVENTS + ("single_dp_tasks", "single_hits", "rescue_hits", "pair_hits", "entries")},
            "per_pair_source_chunk": {k: src.get(k, 0) / (max(1, src.get("chunks", 1)) * lp) for k in NT_EVENTS + ("single_dp_tasks", "single_hits", "rescue_hits", "pair_hits", "entries")},
            "top_rule_across_chunks": {"top_percent": TOP_FLOOR, "failed_pairs_per_other_chunk": o.get("failed_pairs", 0) / n_o, "rescue_rounds_left_out_per_other_chunk": o.get("floor_pairs", 0) / n_o,
                                       "share_of_failed_pairs_left_out": (o.get("floor_pairs", 0) / o["failed_pairs"]) if o.get("failed_pairs") else None,
                                       "instances_left_out_per_pair_per_other_chunk": o.get("floor_tasks", 0) / (n_o * lp),
                                       "share_left_out_by_the_bound_on_the_rescued_mate": (o.get("floor_mate_pairs", 0) / o["failed_pairs"]) if o.get("failed_pairs") else None,
                                       "share_looked_at_by_that_bound": (o.get("floor_mate_tried", 0) / o["failed_pairs"]) if o.get("failed_pairs") else None,
                                       "note": "kernels_floor.hip: the rescue rounds run after the deep-DP rounds of all chunks; a pair whose rescue work in a chunk cannot produce an "
                                               "entry of top% of its best known score loses it there (decided per pair from sound bounds: blocks of five read bases absent from "
                                               "every single-end window; for the pairs that leaves, blocks of six mate bases absent from the text a rescue could search -- "
                                               "MP_TOP_FLOOR_MATE=0: without the second); 0 = off (MP_TOP_FLOOR=0: one pass)"},
            "rescue_certificate_per_pair_per_chunk": dict(zip(("certified_no_alignment", "score_of_10_elsewhere", "bound_reaches_cutoff", "not_eligible"),
                                                              [(b - a) / (S * lp * n_chunks) for a, b in zip(th_a, th_b)])),
            "hit_count_allreduce": {"references": int(len(ref_counts)), "entries_counted": int(ref_counts.sum()), "ranks": world, "backend": (backend if world > 1 else None),
                                    "note": "per-(chunk, reference) counts of the last step's reduced (read, reference, best score) entries, summed over the ranks: the dense vector "
                                            "cc/reassign's pass-1 counters are all-reduced as (megapath_amd/taxcount.py)"},
        }
        # the replica makes room (the PMC children build their own small one; the CPU runs need host memory and the files only)
        lkt_a = lkt_b = None
        if world == 1 and not args.no_cpu_baseline:
            lkt_a, lkt_b = rep.chunks[0].lkt_table(), rep.chunks[1].lkt_table()
        rep.close()
        torch.cuda.empty_cache()
        tmpdir = tempfile.mkdtemp(prefix="mpbench_")
        td_cpu = None
        try:
            if world == 1 and not args.no_pmc:
                try:
                    tr = live_traffic(child_args_base + ["--pmc-child", "nt"], tmpdir, SEED_STAGE_KERNELS)
                except Exception as e:  # noqa: BLE001
                    log("[bench] PMC passes failed: %r" % (e,))
                    tr = None
                if tr:
                    rf = out["roofline"]
                    rf["traffic"] = tr["bytes_per_launch"]
                    t_launch = rf["other_chunks"]["launch_ms"] / 1e3
                    rf["traffic_gbs"] = tr["bytes_per_launch"] / t_launch / 1e9
                    rf["traffic_frac"] = rf["traffic_gbs"] / HBM_PEAK_GBS
                    rf["frac_of_gather_ceiling"] = rf["traffic_gbs"] / gather_gbs
                    rf["traffic_over_requested"] = tr["bytes_per_launch"] / rf["other_chunks"]["bytes_per_launch"]
                    rf["traffic_detail"] = dict(tr, launch="the seed-stage kernels' last launch of the child's step = a chunk none of the reads comes from; compared with other_chunks' bytes and time")
            if world == 1 and not args.no_cpu_baseline:
                cpu_nt, td_cpu = nt_cpu_baseline(args, index_builder, rep, args.read_len, args.nt_cpu_pairs, lkt_b)
                if cpu_nt:
                    out["cpu_baseline"] = dict(cpu_nt, value=cpu_nt["pairs_per_s_one_chunk"] / n_chunks, unit="pairs/s", cores=cpu_nt["threads"], host_cores=os.cpu_count(),
                                               sample="%d pairs of the same read mix, real soap4 (oracle/_ref) -T %d with the NT stage's command line against ONE chunk (the unrelated genome), "
                                                      "alignment phase only; the replica's %d chunks are %d such runs for it (runMegaPath.sh:191-227), so value = that rate / %d"
                                                      % (cpu_nt["pairs"], cpu_nt["threads"], n_chunks, n_chunks, n_chunks))
                else:
                    out["cpu_baseline"] = None
                try:
                    ps = nt_parity_sample(args, index_builder, rep, td_cpu, local_rank, args.read_len, args.nt_cpu_pairs, lkt_a)
                except Exception as e:  # noqa: BLE001
                    log("[bench] NT parity sample failed: %r" % (e,))
                    ps = {"records_equal": None, "first_diff": repr(e)}
                out["parity_sample"] = ps
                if ps and ps.get("records_equal") is False:
                    parity_ok = False
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)
            if td_cpu:
                shutil.rmtree(td_cpu, ignore_errors=True)
    # ---------------------------------------------------------------- legs (one GPU only): configs[1], the other read shapes, the executable
    text, arrs = rep.text_a, rep.arrs_a
    rep.close()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_legs:
        legs = {}
        if want_leg("human_filter"):
            try:
                leg, sub, ok = human_filter_leg(args, H, index_builder, simulate, text, arrs, dev, local_rank, rank, n_bases, want_leg, child_args_base)
                legs["human_filter"] = leg
                legs.update(sub)
                parity_ok = parity_ok and ok
            except Exception as e:  # noqa: BLE001
                legs["human_filter"] = {"error": repr(e)}
                log("[bench] human-filter leg failed: %r" % (e,))
        out["legs"] = legs
    if rank == 0:
        out["parity_ok"] = parity_ok
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and not parity_ok:
        log("[bench] A PARITY SAMPLE DIFFERS FROM THE REFERENCE soap4: see parity_sample / legs.human_filter.parity_sample in the line above")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
