"""One read-shape leg of bench.py alone (for rocprofv3 --kernel-trace --stats): python3 scripts/r4_leg_prof.py C5_mixed_rescue [thermo level]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from megapath_amd import hipapi as H, index_builder, simulate  # noqa: E402

leg = sys.argv[1]
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sims = {"C3_divergent": dict(read_len=150, sub_max=0.08, random_frac=0.4),
        "C4_2x250": dict(read_len=250, frag_mean=450.0, frag_sd=50.0, frag_max=700),
        "C5_mixed_rescue": dict(read_len=300, len_range=(75, 300), broken_mate_frac=0.3, frag_mean=450.0, frag_sd=50.0, frag_max=700)}
dev = torch.device("cuda", 0)
n = int(float(os.environ.get("MP_BENCH_GENOME_MBP", "2950")) * 1e6)
text = simulate.genome(n, seed=3, device=dev)
b = np.linspace(0, n, 25).astype(np.int64)
tc = text.cpu().numpy()
arrs = index_builder.build_index_arrays([("chr%d" % (i + 1), tc[b[i]:b[i + 1]]) for i in range(24)], sa_interval=16, device=dev)
del tc
torch.cuda.empty_cache()
gix = H.Index.from_arrays(0, **arrs)
if leg == "human_filter":
    lw = bench.Workload(text, 1 << 20, seed=1000, read_len=150, L=150)
    par = H.AlignParams(lw.max_len, 750, 130, 130, H.MmpParams.human())
else:
    lw = bench.Workload(text, 1 << 18, seed=2000, **sims[leg])
    par = H.AlignParams(lw.max_len, 750, 130, 130, H.MmpParams.nt2())
ctx = H.Context(gix)
ctx.set_dp_thermo(level)
bench.step(ctx, lw, par)
el, la = bench.run_steps(ctx, lw, par, 4)
print(leg, "thermo level", level, "pairs/s", round(4 * lw.n_pairs / el), {k[3:]: round(la[k] / 4, 2) for k in la if k.startswith("ms_")}, "thermo stats", ctx.dp_thermo_stats())
ctx.close()
gix.close()
