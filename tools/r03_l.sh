#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so
for d in 0 1 2 4 6 7 8 16 33 39; do PP_XA_DBG=$d timeout 120 python tools/xattn_ablate.py 2>&1 | grep PP_XA; done | tee $O/ablate.txt
