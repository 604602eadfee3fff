#!/bin/bash
# round 3, call B: register-direct GEMM epilogue -- parity suites, then A/B against the staged build (same box)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $O/ops.log 2>&1; echo "ops rc=$?"; tail -4 $O/ops.log
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_real_shapes_gpu.py -m gpu -q -s -p no:cacheprovider --timeout=600 > $O/models.log 2>&1; echo "models rc=$?"; tail -4 $O/models.log; grep "real-shape parity" $O/models.log
timeout 600 python -m pytest tests/test_config_parity_gpu.py -q -s -k "teacher" -p no:cacheprovider --timeout=600 > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
timeout 300 python -m pytest tests/test_golden.py -m gpu -q -k "strength" -p no:cacheprovider > $O/golden.log 2>&1; echo "golden rc=$?"; tail -3 $O/golden.log
for rep in 1 2; do
  for v in staged direct; do
    if [ $v = staged ]; then export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_staged.so; else unset PP_LAB PP_LIB; fi
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
  done
done
unset PP_LAB PP_LIB
for v in staged direct; do
  if [ $v = staged ]; then export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_staged.so; else unset PP_LAB PP_LIB; fi
  timeout 400 python tools/gemm_pp_bench.py --tiles 53 --no-cold --rounds 2 --json $O/gemm_$v.json > $O/gemm_$v.txt 2>&1; echo "gemm bench $v rc=$?"
done
unset PP_LAB PP_LIB
python - <<'PY'
import json
a={ (tuple(sorted(r['sig'].items())),r['tile'],r['splitk']):r for r in json.load(open('gpurun_out/r03b/gemm_staged.json'))}
b={ (tuple(sorted(r['sig'].items())),r['tile'],r['splitk']):r for r in json.load(open('gpurun_out/r03b/gemm_direct.json'))}
ts=td=0
for k in a:
    if k[1]!=0 or k not in b: continue
    s=a[k]['sig']; n=a[k]['count']
    ts+=a[k]['hot_us']*n; td+=b[k]['hot_us']*n
    print(f"{'conv' if s['x_mode'] else 'lin '} M{s['M']:6d} N{s['N']:6d} K{s['K']:6d} x{n:2d} act{s['act']} vt{int(s['vt'])}  staged {a[k]['hot_us']:7.1f}  direct {b[k]['hot_us']:7.1f}  d {b[k]['hot_us']-a[k]['hot_us']:+6.1f}")
print('per step hot sum: staged', round(ts,1), 'direct', round(td,1))
PY
cat gpurun_out/parity_r03.txt 2>/dev/null | tail -16
