#!/bin/bash
# attention: 64 queries per wave on 32-key tiles (PP_ATTN_QB=2) against the shipped 32 x 64 configuration
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02j
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -p no:cacheprovider -k "attention" > $O/t1.log 2>&1; echo "default tests rc=$?"; tail -2 $O/t1.log
PP_ATTN_QB=2 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -p no:cacheprovider -k "attention" > $O/t2.log 2>&1; echo "QB=2 tests rc=$?"; tail -4 $O/t2.log
for i in 1 2 3; do
  for v in 1 2; do echo -n "QB=$v "; PP_ATTN_QB=$v timeout 120 python tools/attn_ablate.py one; done
done 2>&1 | tee $O/ab.txt
for i in 1 2; do for v in 1 2; do
  PP_ATTN_QB=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QB=$v step', round(d['ms_per_denoise_step'],3))"
done; done | tee -a $O/ab.txt
