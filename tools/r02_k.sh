#!/bin/bash
# 64-query attention waves (PP_ATTN_QB=2) with 4- and 8-wave workgroups against the 32-query kernel
set -u
cd "$(dirname "$0")/.."
PP_ATTN_QB=2 PP_ATTN_NW=8 timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "attention" 2>&1 | tail -1
for i in 1 2 3; do
  echo -n "QB=1 NW=4 "; timeout 120 python tools/attn_ablate.py one
  echo -n "QB=2 NW=4 "; PP_ATTN_QB=2 timeout 120 python tools/attn_ablate.py one
  echo -n "QB=2 NW=8 "; PP_ATTN_QB=2 PP_ATTN_NW=8 timeout 120 python tools/attn_ablate.py one
done 2>&1
for i in 1 2; do for v in "1 4" "2 4" "2 8"; do set -- $v
  PP_ATTN_QB=$1 PP_ATTN_NW=$2 timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QB=$1 NW=$2 step', round(d['ms_per_denoise_step'],3))"
done; done
