#!/bin/bash
# round 3, call M: fused cross-attention block inside the UNet (lab opt-in): network parity + step A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp PP_LAB=1
PP_XATTN_FUSED=1 timeout 600 python -m pytest tests/test_real_shapes_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -x -k "unet" > $O/t.log 2>&1; echo "tests rc=$?"; tail -5 $O/t.log
for rep in 1 2; do for f in 0 1; do
PP_XATTN_FUSED=$f timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED=$f rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3), 'launches', d.get('launches_per_step'))"
done; done | tee $O/ab.txt
