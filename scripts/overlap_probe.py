"""How much does running two half-batches concurrently on two contexts (two HIP streams, one index) buy over one context
with the whole batch?  The seeding of one half overlaps with the DP of the other.  Run on the GPU box:
    python scripts/overlap_probe.py [--genome-mbp 2950] [--steps 4]"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megapath_amd import hipapi as H  # noqa: E402
from megapath_amd import index_builder, simulate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genome-mbp", type=float, default=2950)
ap.add_argument("--pairs", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--contexts", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n = int(args.genome_mbp * 1e6)
text = simulate.genome(n, seed=7, device=dev)
cuts = [n * i // 24 for i in range(25)]
tc = text.cpu().numpy()
arrs = index_builder.build_index_arrays([("chr%d" % (i + 1), tc[cuts[i]:cuts[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
gix = H.Index.from_arrays(0, **arrs)
del arrs
torch.cuda.empty_cache()
P, L = args.pairs, 150
W = (L + 1 + 15) // 16
codes, lens = simulate.paired_reads(text, P, L, seed=1000)
packed = simulate.pack_codes(codes, W)
lens_i = lens.contiguous()
params = H.AlignParams(L, 750, 130, 130, H.MmpParams.nt2())


def run(n_ctx):
    per = P // n_ctx
    ctxs = [H.Context(gix) for _ in range(n_ctx)]

    def step(i):
        c = ctxs[i]
        c.set_reads_device(packed.data_ptr() + i * 2 * per * W * 4, W, lens_i.data_ptr() + i * 2 * per * 4, 2 * per, L, keep=(packed, lens_i))
        c.deep_dp_round(params, fetch=False)
        c.rescue_round(params, False, fetch=False)
        c.pair_hits_view()

    def worker(i, k, delay):
        time.sleep(delay)
        for _ in range(k):
            step(i)
    for i in range(n_ctx):
        step(i)                                             # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, args.steps, 0.02 * i)) for i in range(n_ctx)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return args.steps * P / dt, dt / args.steps * 1e3


a = run(1)
print("1 context : %.2f M pairs/s, %.1f ms per %d pairs" % (a[0] / 1e6, a[1], P), flush=True)
b = run(args.contexts)
print("%d contexts: %.2f M pairs/s, %.1f ms per %d pairs (%.0f %% more)" % (args.contexts, b[0] / 1e6, b[1], P, 100 * (b[0] / a[0] - 1)), flush=True)
