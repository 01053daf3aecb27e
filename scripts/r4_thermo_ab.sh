#!/bin/bash
set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python3 scripts/r4_thermo_ab.py gpurun_out/r4/cert_A.npz > gpurun_out/r4/cert_A.log 2>&1 || { tail -20 gpurun_out/r4/cert_A.log; exit 1; }
cp megapath_amd/libmegapath_hip_ab.so megapath_amd/libmegapath_hip.so
timeout -k 10 300 python3 scripts/r4_thermo_ab.py gpurun_out/r4/cert_B.npz > gpurun_out/r4/cert_B.log 2>&1 || { tail -20 gpurun_out/r4/cert_B.log; exit 1; }
python3 - <<'PY'
import numpy as np
a=np.load("gpurun_out/r4/cert_A.npz"); b=np.load("gpurun_out/r4/cert_B.npz")
for name in ("narrow","short","wide","rescue","rescue_wide"):
    A=[a["%s_%d"%(name,r)] for r in range(3)]; B=[b["%s_%d"%(name,r)] for r in range(3)]
    print(name, "A reps equal", all((x==A[0]).all() for x in A), "B reps equal", all((x==B[0]).all() for x in B), "A0 vs B0 differ", int((A[0]!=B[0]).sum()), "A certified", int(A[0].sum()), "B", int(B[0].sum()))
    d=np.nonzero(A[0]!=B[0])[0][:12]
    if len(d): print("   ", [(int(t), int(A[0][t]), int(B[0][t]), int(a[name+"_kind"][t]), int(a[name+"_len"][t]), int(a[name+"_m"][t])) for t in d])
PY
