"""End-to-end throughput of the soap4-compatible executable from FASTQ files to annotated FASTQ (and BAM with --bam):
what a pipeline maintainer sees, host parsing and printing included.  Run on the GPU box:
    python scripts/e2e_cli.py [--mbp 256] [--pairs 1048576] [--bam] [--gz]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megapath_amd import index_builder, simulate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=int, default=256)
ap.add_argument("--pairs", type=int, default=1 << 20)
ap.add_argument("--bam", action="store_true")
ap.add_argument("--gz", action="store_true")
ap.add_argument("--threads", type=int, default=2)
ap.add_argument("--null", action="store_true", help="stdout to /dev/null: the host pipeline without the file system")
ap.add_argument("--extra", default="", help="extra soap4 flags, e.g. -lsam-noseq")
args = ap.parse_args()
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
n = args.mbp << 20
t0 = time.time()
text = simulate.genome(n, seed=3, device=dev)
n_seq = 24
cuts = [n * i // n_seq for i in range(n_seq + 1)]
tc = text.cpu().numpy()
seqs = [("chr%d" % (i + 1), tc[cuts[i]:cuts[i + 1]]) for i in range(n_seq)]
arrs = index_builder.build_index_arrays(seqs, sa_interval=16, device=dev)
prefix = os.path.join(tmp, "ref.index")
index_builder.write_index_files(prefix, arrs)
print("index %d Mbp built and written in %.1f s" % (args.mbp, time.time() - t0), flush=True)
t0 = time.time()
codes, lens = simulate.paired_reads(text, args.pairs, 150, seed=1000)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
seq = lut[codes.long()].cpu().numpy()                         # (2P, 150) ASCII
P = args.pairs
for mate in (0, 1):
    rows = seq[mate::2]
    names = np.char.add(np.char.add("@r", np.arange(P).astype(str)), "/%d" % (mate + 1)).astype("S")
    with open(os.path.join(tmp, "r_%d.fq" % (mate + 1)), "wb") as f:
        chunk = 1 << 16
        tail = b"\n+\n" + b"I" * 150 + b"\n"
        for off in range(0, P, chunk):
            f.write(b"".join(nm + b"\n" + r.tobytes() + tail for nm, r in zip(names[off:off + chunk], rows[off:off + chunk])))
f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
if args.gz:
    subprocess.check_call(["gzip", "-1", f1, f2])
    f1, f2 = f1 + ".gz", f2 + ".gz"
print("reads written in %.1f s" % (time.time() - t0), flush=True)
del text, codes
torch.cuda.empty_cache()
cmd = [os.path.join(ROOT, "bin", "soap4-mi355x"), "pair", prefix, f1, f2, "-o", os.path.join(tmp, "out"), "-C", os.path.join(ROOT, "tests", "soap4-nt2.ini"),
       "-L", "151", "-T", str(args.threads), "-u", "750", "-F", "-nc"] + (["-b"] if args.bam else []) + args.extra.split()
t0 = time.time()
with open("/dev/null" if args.null else os.path.join(tmp, "out.fq"), "wb") as fo:
    r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, MP_TIMING="1"))
wall = time.time() - t0
print(r.stderr.decode()[-1500:])
print("rc %d, wall %.2f s (index load included) -> %.0f pairs/s over the whole process" % (r.returncode, wall, P / wall))
print("stdout bytes", 0 if args.null else os.path.getsize(os.path.join(tmp, "out.fq")), [(f, os.path.getsize(os.path.join(tmp, f))) for f in os.listdir(tmp) if f.startswith("out.")])
