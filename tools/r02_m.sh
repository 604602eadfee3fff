#!/bin/bash
# A/B of one attention switch given as $1 (PP_ATTN_EARLY, PP_ATTN_PAIR: experiment builds only, both reverted --
# profiles/r02_rejected_experiments.txt, item 13; PP_ATTN_QB / PP_ATTN_NW exist in the product)
set -u
cd "$(dirname "$0")/.."
V=${1:-PP_ATTN_EARLY}
env $V=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "attention" 2>&1 | tail -1
for i in 1 2 3; do
  for v in 0 1; do echo -n "$V=$v "; env $V=$v timeout 120 python tools/attn_ablate.py one; done
done 2>&1
for i in 1 2; do for v in 0 1; do
  env $V=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$v step', round(d['ms_per_denoise_step'],3))"
done; done
