#!/bin/bash
# certificates instance by instance, then the NT step, under the library as built (A) and under megapath_amd/libmegapath_hip_ab.so (B)
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4
cp megapath_amd/libmegapath_hip.so /tmp/lib_a.so
bash scripts/r4_thermo_ab.sh
cp /tmp/lib_a.so megapath_amd/libmegapath_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_thermo.py tests/test_gpu_top_floor.py -x -q 2>&1 | tail -3
bash scripts/r4_ab.sh
