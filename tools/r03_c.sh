#!/bin/bash
# round 3, call C: write-through (sc1) output stores against plain stores, same box
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $O/ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/ops.log
timeout 900 python -m pytest tests/test_models_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $O/models.log 2>&1; echo "models rc=$?"; tail -3 $O/models.log
for rep in 1 2 3; do
  for v in plain wt; do
    if [ $v = plain ]; then export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_plainst.so; else unset PP_LAB PP_LIB; fi
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
for v in plain wt; do
  if [ $v = plain ]; then export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_plainst.so; else unset PP_LAB PP_LIB; fi
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
r=d['roofline']; h=d['hbm_roofline']; m=d['mfma_families_live']
print('$v', 'step', round(d['ms_per_denoise_step'],3), 'conv us/launch', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4),
      '| lin', h['linear + conv1x1 (plain GEMMs)']['avg_launch_us'], 'gn', h['groupnorm_apply']['avg_launch_us'],
      '| attn', m['attention']['us_per_step'], 'geglu', m['linear_geglu']['us_per_step'])
PY
done
unset PP_LAB PP_LIB
