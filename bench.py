#!/usr/bin/env python3
"""bench.py -- paired 150 bp reads aligned per second against a 2BWT index on MI355X (BASELINE.json metric).

One process per GPU (`torchrun`-style env: RANK / LOCAL_RANK / WORLD_SIZE).  Reads shard across ranks, the index is
replicated into every GPU's HBM, no data-path collective: `"scaling": "weak"`.  A *step* is one
pass of the hot path over one batch of synthetic pairs whose packed reads already sit in HBM.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import ctypes as C
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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(args, text_cpu, arrs, sample_codes, tmpdir, lkt_table=None):
    """Times the CPU path on a bounded sample of the same workload, on the host cores of this box.
    kind = "reference": the real soap4 built by oracle/Makefile.ref (oracle/_ref/soap4 travels with the snapshot);
    kind = "port": the plain-C oracle when that binary is absent."""
    from megapath_amd import index_builder, simulate
    n_pairs = sample_codes.shape[0] // 2
    fq1, fq2 = os.path.join(tmpdir, "s_1.fq"), os.path.join(tmpdir, "s_2.fq")
    codes = sample_codes.cpu().numpy()
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    qual = b"I" * codes.shape[1]
    with open(fq1, "wb") as f1, open(fq2, "wb") as f2:
        for p in range(n_pairs):
            f1.write(b"@r%d/1\n" % p + lut[codes[2 * p]].tobytes() + b"\n+\n" + qual + b"\n")
            f2.write(b"@r%d/2\n" % p + lut[codes[2 * p + 1]].tobytes() + b"\n+\n" + qual + b"\n")
    prefix = os.path.join(tmpdir, "ref.index")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "soap4")
    use_ref = os.path.exists(ref_bin)
    index_builder.write_index_files(prefix, arrs, write_lkt=use_ref, lkt_table=lkt_table if use_ref else None)
    cores = min(os.cpu_count() or 1, 16)
    if use_ref:
        ini = os.path.join(tmpdir, "soap4-nt2.ini")
        with open(ini, "w") as f:
            f.write(NT2_INI)
        cmd = [ref_bin, "pair", prefix, fq1, fq2, "-o", os.path.join(tmpdir, "out"), "-C", ini, "-L", str(args.read_len + 1),
               "-T", str(cores), "-u", "750", "-F", "-nc"]
        t0 = time.time()
        with open(os.path.join(tmpdir, "ref.out"), "wb") as fo, open(os.path.join(tmpdir, "ref.err"), "wb") as fe:
            rc = subprocess.call(cmd, stdout=fo, stderr=fe)
        wall = time.time() - t0
        align_s = None
        for line in open(os.path.join(tmpdir, "ref.err"), errors="replace"):
            if "Overall alignment time" in line:
                align_s = float(line.split(":")[1].split()[0])
        if rc != 0 or not align_s:
            log("[bench] reference soap4 failed (rc=%d); wall %.1fs" % (rc, wall))
            return None
        return {"value": n_pairs / align_s, "unit": "pairs/s", "cores": cores, "kind": "reference",
                "sample": "%d pairs of the same workload, real soap4 (-T %d), alignment phase only (index/read load excluded)" % (n_pairs, cores)}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    lib = O.oracle()
    ix = lib.mpo_index_open(prefix.encode())
    p = O.default_params(L=args.read_len + 1)
    st = O.AlignStats()
    t0 = time.time()
    lib.mpo_align_fastq(ix, C.byref(p), fq1.encode(), fq2.encode(), os.path.join(tmpdir, "o.fq").encode(), None, C.byref(st))
    dt = time.time() - t0
    return {"value": n_pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d pairs of the same workload, scalar C oracle, whole path" % n_pairs}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("MP_BENCH_GENOME_MBP", "2950")))
    ap.add_argument("--pairs-per-step", type=int, default=int(os.environ.get("MP_BENCH_PAIRS", str(1 << 20))))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample-pairs", type=int, default=int(os.environ.get("MP_BENCH_CPU_PAIRS", "100000")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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

    from megapath_amd import hipapi as H
    from megapath_amd import index_builder, simulate

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
        log("[bench] genome %.0f Mbp: suffix sort + BWT %.1fs, HBM re-layout + upload %.1fs, index %.2f GB in HBM"
            % (args.genome_mbp, t1 - t0, t2 - t1, gix.info.device_bytes / 1e9))
    torch.cuda.empty_cache()
    g_ms, g_bytes = gix.bench_gather64()
    gather_gbs = g_bytes / (g_ms * 1e-3) / 1e9
    if rank == 0:
        log("[bench] random 64-B gather ceiling over the rank blocks: %.0f GB/s" % gather_gbs)

    P = args.pairs_per_step
    W = (args.read_len + 1 + 15) // 16
    codes, lens = simulate.paired_reads(text, P, args.read_len, seed=1000 + rank)
    packed = simulate.pack_codes(codes, W)
    lens_i = lens.contiguous()
    ctx = H.Context(gix)
    ctx.set_reads_device(packed.data_ptr(), W, lens_i.data_ptr(), 2 * P, args.read_len, keep=(packed, lens_i))
    params = H.AlignParams(args.read_len, 750, 130, 130, H.MmpParams.nt2())
    lib = H.load_library()

    def one_step():
        # a new batch arrives: forget the seeds of the previous one, then run the whole path on the device:
        # deep-DP round (seed, locate, merge, pair, both DP legs) and the rescue passes of the pairs it left unaligned
        ctx.set_reads_device(packed.data_ptr(), W, lens_i.data_ptr(), 2 * P, args.read_len, keep=(packed, lens_i))
        _, nh, st = ctx.deep_dp_round(params, fetch=False)             # the hits start travelling to the host ...
        _, _, rst = ctx.rescue_round(params, False, fetch=False)       # ... while the rescue passes run
        hits, nh = ctx.pair_hits_view()                                # the step ends when they have arrived
        return hits, nh, st, rst

    for _ in range(args.warmup):
        one_step()
    # Event counts of the reference algorithm for this batch (every table look-up and rank query soap4 itself issues):
    # one untimed step with the index's 17-mer pre-filter ignored.  SURVEY 8d prices the seed stage with these.
    ref_events = None
    if gix.info.prefilter_k:
        ctx.set_prefilter(False)
        _, _, st0, _ = one_step()
        ref_events = dict(occ=int(st0.seed.n_occ), lkt=int(st0.seed.n_lkt), sa=int(st0.seed.n_sa), seedpos=int(st0.seed.n_seedpos))
        ctx.set_prefilter(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tstart = time.perf_counter()
    acc = dict(seed=0.0, locate=0.0, merge=0.0, pair=0.0, fill=0.0, trace=0.0, occ=0, lkt=0, sa=0, seedpos=0, kmer=0, cand=0, hits=0,
               rescue_fill=0.0, rescue_trace=0.0, rescue_tasks=0.0, failed=0, stasks=0, rtasks=0)
    band0 = ctx.dp_band_stats()
    for _ in range(args.steps):
        hits, nh, st, rst = one_step()
        acc["rescue_fill"] += rst.ms_dp_fill; acc["rescue_trace"] += rst.ms_dp_trace; acc["rescue_tasks"] += rst.ms_tasks
        acc["failed"] += rst.n_failed_pairs; acc["stasks"] += rst.n_single_tasks; acc["rtasks"] += rst.n_rescue_tasks
        acc["seed"] += st.seed.ms_seed; acc["locate"] += st.seed.ms_locate; acc["merge"] += st.seed.ms_merge
        acc["pair"] += st.ms_pair; acc["fill"] += st.ms_dp_fill; acc["trace"] += st.ms_dp_trace
        acc["occ"] += st.seed.n_occ; acc["lkt"] += st.seed.n_lkt; acc["sa"] += st.seed.n_sa; acc["seedpos"] += st.seed.n_seedpos
        acc["kmer"] += st.seed.n_prefilter; acc["blk"] = acc.get("blk", 0) + st.seed.n_occ_loaded
        acc["cand"] += st.n_candidates; acc["hits"] += st.n_pair_hits
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - tstart
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    band1 = ctx.dp_band_stats()
    if rank == 0:
        K = args.steps
        total_pairs = world * K * P
        # ---- roofline of the seed stage (K1 + K2): algorithmic bytes from the event counters the kernels emit
        # B_seed = 64*N_occ + 16*N_lkt + 8*N_sa + 0.25*sum(read_len) + 16*N_seedpos   
        # with the events of the reference algorithm (every step processes the same resident batch); what the kernels
        # actually issued (fewer look-ups when the 17-mer pre-filter is resident, 4 B per bit probe) is reported next to it
        ev = ref_events if ref_events else dict(occ=acc["occ"] / K, lkt=acc["lkt"] / K, sa=acc["sa"] / K, seedpos=acc["seedpos"] / K)
        b_seed = K * (64 * ev["occ"] + 16 * ev["lkt"] + 8 * ev["sa"] + 0.25 * (2 * P * args.read_len) + 16 * ev["seedpos"])
        b_exec = 64 * acc["occ"] + 16 * acc["lkt"] + 8 * acc["sa"] + 0.25 * (2 * P * K * args.read_len) + 16 * acc["seedpos"] + 4 * acc["kmer"]
        t_seed_s = (acc["seed"] + acc["locate"]) / 1e3
        # `achieved` is priced with the events the kernels themselves performed and counted ("counters the kernel emits",
        # N_occ = 2 per backward-extension step performed -- whether the step loaded rank blocks or was taken by the 16-mer table or by
        # comparing with the text; starts the 17-mer pre-filter rejects perform no steps).  The rank blocks actually loaded are in
        # events_per_pair.rank_blocks_loaded; the rate at which the reference algorithm's look-ups are delivered is reported next to it
        achieved = b_exec / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
        achieved_ref = b_seed / t_seed_s / 1e9 if t_seed_s > 0 else 0.0
        kernel_ms = {k: acc[k] / K for k in ("seed", "locate", "merge", "pair", "fill", "trace", "rescue_tasks", "rescue_fill", "rescue_trace")}
        n_inst = 2 * acc["cand"] + acc["stasks"] + acc["rtasks"]
        cells = (2 * acc["cand"] + acc["stasks"]) * (args.read_len + 61) * args.read_len + acc["rtasks"] * (750 - args.read_len + args.read_len + 1) * args.read_len
        out = {
            "metric": "paired 150bp reads aligned/sec vs 2BWT index",
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "2x%dbp synthetic pairs vs synthetic %.0f Mbp genome in 24 sequences (human-filter stage; GRCh38 indexes ~2.95 Gbp of non-N bases), %d pairs/step/GPU, "
                                   "whole path on the device: deep-DP round (MMP seeding, pairing, both DP legs + traceback) and rescue passes (single-end DP, mate rescue)" % (args.read_len, args.genome_mbp, P),
                       "index_replicated_per_gpu": True, "pairs_per_step_per_gpu": P, "genome_mbp": args.genome_mbp,
                       "ini": "soap4-nt2.ini", "L": args.read_len + 1, "u": 750},
            "roofline": {"bound": "hbm", "kernel": "seed_kernel+locate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "random64_gather_ceiling_gbs": gather_gbs, "frac_of_gather_ceiling": achieved / gather_gbs,
                         "algorithmic_bytes_per_pair": b_exec / (K * P), "algorithmic_bytes_per_launch": b_exec / K,
                         "launch_ms": (acc["seed"] + acc["locate"]) / K,
                         "reference_algorithm": {"bytes_per_pair": b_seed / (K * P), "bytes_per_launch": b_seed / K, "delivered_gbs": achieved_ref,
                                                 "frac_of_peak": achieved_ref / HBM_PEAK_GBS,
                                                 "note": "look-ups soap4 itself issues for this batch (one untimed step with the 17-mer pre-filter "
                                                         "and the dense SA samples ignored); the timed kernels skip most of them"},
                         "prefilter_k": int(gix.info.prefilter_k)},
            "kernel_ms_per_step": kernel_ms,
            "dp": {"bound": "valu", "unit": "Gcells/s", "achieved": cells / ((acc["fill"] + acc["rescue_fill"]) / 1e3) / 1e9 if acc["fill"] else 0.0,
                   "instances_per_step": n_inst / K,
                   "band_instances_per_step": (band1[0] - band0[0]) / K, "band_left_to_full_table_per_step": (band1[1] - band0[1]) / K,
                   "note": "integer max/add recurrences: VALU-bound, no MFMA, not an HBM roofline (SURVEY 8d).  `achieved` counts the "
                           "full-table cells of every instance (SURVEY 8d) over the DP fill time; the deep-DP legs run exact banded kernels (a "
                           "block-match pre-filter confines each alignment to 27 of its window's ~211 diagonals, perfect matches need no "
                           "table) and hand the instances they cannot decide to the full-table kernels; the rescue batches go through a "
                           "score-only filter first"},
            "events_per_pair": {"occ": ev["occ"] / P, "lkt": ev["lkt"] / P, "sa": ev["sa"] / P,
                                "executed_occ": acc["occ"] / (K * P), "rank_blocks_loaded": acc.get("blk", 0) / (K * P), "executed_lkt": acc["lkt"] / (K * P),
                                "prefilter_probes": acc["kmer"] / (K * P),
                                "candidates": acc["cand"] / (K * P), "pair_hits": acc["hits"] / (K * P),
                                "failed_pairs": acc["failed"] / (K * P), "single_dp_tasks": acc["stasks"] / (K * P), "rescue_dp_tasks": acc["rtasks"] / (K * P)},
        }
        # HBM traffic of the seed stage from the committed PMC passes (separate rocprofv3 --pmc runs of this very command,
        # profiles/r01_pmc_seed_stage.json); only quoted when the workload is the one those passes measured
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_seed_stage.json")))
            wl = pmc["workload"]
            if wl["genome_mbp"] == args.genome_mbp and wl["pairs_per_step"] == P and wl["read_len"] == args.read_len:
                out["roofline"]["traffic"] = pmc["seed_stage_hbm_bytes_per_launch"]
                # VALU occupancy of the DP kernels from the committed SQ-counter pass of this command (profiles/r01_pmc_dp_valu.json)
                valu = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_dp_valu.json")))["per_kernel"]
                out["dp"]["valu_busy_frac"] = {k.replace("void ", ""): round(v["valu_busy_frac"], 3) for k, v in valu.items()}
        except Exception:  # noqa: BLE001
            pass
        cb = None
        if world == 1 and not args.no_cpu_baseline:
            tmpdir = tempfile.mkdtemp(prefix="mpbench_")
            try:
                cb = cpu_baseline(args, text_cpu, arrs, codes[: 2 * min(args.cpu_sample_pairs, P)], tmpdir, lkt_table=gix.lkt_table())
            except Exception as e:  # noqa: BLE001 -- the baseline is auxiliary; report its absence, never hide the GPU number
                log("[bench] cpu baseline failed: %r" % (e,))
            finally:
                shutil.rmtree(tmpdir, ignore_errors=True)
        out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    ctx.close()
    gix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
