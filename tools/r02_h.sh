#!/bin/bash
# attention: 8-wave blocks (one per CU) against 4-wave blocks (two per CU): parity of the 8-wave variant, A/B of the kernel
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02h
mkdir -p $O
PP_ATTN_NW=8 timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "attention" > $O/t8.log 2>&1; echo "nw8 tests rc=$?"; tail -2 $O/t8.log
for i in 1 2; do
  for nw in 4 8; do echo -n "NW=$nw: "; PP_ATTN_NW=$nw timeout 120 python tools/attn_ablate.py one; done
done 2>&1 | tee $O/ab.txt
for nw in 4 8; do
  PP_ATTN_NW=$nw timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NW=$nw step', round(d['ms_per_denoise_step'],3))"
done | tee -a $O/ab.txt
