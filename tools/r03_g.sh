#!/bin/bash
# round 3, call G: attention with 64-key LDS tiles (OPT 16) / delayed cvt (OPT 8) against the shipping OPT 5
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so
timeout 600 python tools/attn_opt_ab.py > $O/attn_opt.txt 2>&1; echo "attn ab rc=$?"; cat $O/attn_opt.txt
for o in 21 29; do
  PP_ATTN_OPT=$o timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_config_parity_gpu.py -m gpu -q -p no:cacheprovider -k "attention" 2>&1 | tail -2
done
for rep in 1 2; do
  for o in 5 21 29 13; do
    PP_ATTN_OPT=$o timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OPT $o rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
