#!/bin/bash
# the 4-wave 256x160 GEMM tile (id 63, experiment build only: profiles/r02_rejected_experiments.txt, item 14)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02n
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "gemm or conv or linear or geglu or layernorm or gn_" -x > $O/t.log 2>&1; echo "tests rc=$?"; tail -5 $O/t.log
python tools/gemm_pp_bench.py --tiles 53,63 --conv-only --min-m 8192 --rounds 2 --no-cold 2>&1 | tee $O/bench.txt | grep -v amdgpu | tail -42
