#!/bin/bash
# round 3, call D: attention loop changes (OPT bits) A/B, then the step with OPT 0 / best
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so
timeout 600 python tools/attn_opt_ab.py > $O/attn_opt.txt 2>&1; echo "attn ab rc=$?"; cat $O/attn_opt.txt
for rep in 1 2; do
  for o in 0 7 1 2 4; do
    PP_ATTN_OPT=$o timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OPT $o rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
unset PP_LAB PP_LIB
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipping lib ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
