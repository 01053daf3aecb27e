"""RCCL executed on the GPU box: the path's one exchange step -- the per-taxon counters of cc/reassign pass 1, and the dense
per-reference hit counts bench.py sums over its ranks -- through torch.distributed's `nccl` backend (= RCCL on ROCm) on DEVICE tensors.

A one-GPU box gives a process group of one rank (RCCL refuses two ranks on one device); the collectives still go through RCCL's
communicator, kernels and stream ordering, which is what had never run before round 4 (both multi-rank rehearsals used gloo).  The
child is a fresh process (plain subprocess, nothing is exec'ed over a process that has touched the GPU).  The multi-rank logic itself
(shards, union of taxids, merge of the pair table) is covered by the world-size-2 gloo test in test_host_logic.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch
import torch.distributed as dist
from megapath_amd import taxcount
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)            # as bench.py does for N > 1
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g4 = os.path.join(sys.argv[1], "tests", "golden", "g4")
lines = open(os.path.join(g4, "expected.lsam.id"), "rb").read().splitlines(keepends=True)
p = taxcount.pass1_from_lsam_id(lines, 40)
assert p.counts and p.uniq and p.intersect
total = taxcount.all_reduce_pass1(p, device=dev, always=True)      # all_gather (sizes, padded rows) + all_reduce(SUM) on device tensors
assert total == p, (total, p)
want = sorted(l for l in open(os.path.join(g4, "expected.explains")).read().split("\n") if l)
assert sorted("%d explains %d" % e for e in taxcount.explains(total)) == want          # the real reassign's relation, from counters that went through RCCL
assert b"".join(taxcount.apply_explains(lines, taxcount.explains(total))) == open(os.path.join(g4, "expected.ra.lsam.id"), "rb").read()
counts = np.arange(1, 100001, dtype=np.int64) * 7919
got = taxcount.allreduce_counts(counts, dev)                       # bench.py's per-reference hit counts
assert got.dtype == np.int64 and np.array_equal(got, counts)
t = torch.tensor([3.25], device=dev, dtype=torch.float64)         # bench.py's max-over-ranks of the elapsed time
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok", torch.cuda.nccl.version(), torch.cuda.get_device_name(0))
'''


def test_pass1_counters_through_rccl_on_device_tensors(tmp_path):
    w = tmp_path / "rccl_child.py"
    w.write_text(_CHILD)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(w), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rccl ok" in r.stdout, r.stdout
