#!/bin/bash
# round 3, call K: fused cross-attention block -- op parity + A/B against the three-launch chain
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "fused_cross_attention" > $O/t.log 2>&1; echo "tests rc=$?"; tail -12 $O/t.log
timeout 200 python tools/xattn_ab.py 2>&1 | grep "M=" | tee $O/ab.txt
