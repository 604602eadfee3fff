#!/bin/bash
# round 3, call E: (1) attention: VALU-swap row maximum & co. (OPT bits), (2) weight prefetch from the GroupNorm-apply launches
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -m gpu -q -p no:cacheprovider -k "attention or groupnorm or gn" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops.log
timeout 600 python -m pytest tests/test_config_parity_gpu.py -q -p no:cacheprovider -k "attention" > $O/par.log 2>&1; echo "parity rc=$?"; tail -2 $O/par.log
export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so
timeout 600 python tools/attn_opt_ab.py > $O/attn_opt.txt 2>&1; echo "attn ab rc=$?"; cat $O/attn_opt.txt
for rep in 1 2; do
  for o in 0 1 5 7; do
    PP_ATTN_OPT=$o timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OPT $o rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
unset PP_LIB
for rep in 1 2 3; do
  for v in 0 1; do
    PP_LAB=1 PP_WEIGHT_PREFETCH=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch=$v rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
for v in 0 1; do
  PP_LAB=1 PP_WEIGHT_PREFETCH=$v timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_pf$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$O/bench_pf$v.json').read().strip().splitlines()[-1])
r=d['roofline']; h=d['hbm_roofline']; m=d['mfma_families_live']
print('prefetch=$v', 'step', round(d['ms_per_denoise_step'],3), 'conv us/launch', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4),
      '| lin', h['linear + conv1x1 (plain GEMMs)']['avg_launch_us'], 'gn', h['groupnorm_apply']['avg_launch_us'],
      '| attn', m['attention']['us_per_step'], 'geglu', m['linear_geglu']['us_per_step'])
PY
done
