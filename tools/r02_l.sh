#!/bin/bash
# 64-queries-per-wave attention as the default: parity suites, then config 2 / config 5 against PP_ATTN_QB=1
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02l
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_real_shapes_gpu.py tests/test_models_gpu.py -q -p no:cacheprovider -m gpu -k "attention or config2 or transformer or unet or forward or full_size" > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
for v in 1 0; do
  PP_ATTN_QB=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 QB=$v step', round(d['ms_per_denoise_step'],3), round(d['value'],3))"
  PP_ATTN_QB=$v timeout 600 python bench.py --config v2 --latent 128 --per-gpu 2 --denoise-steps 30 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 QB=$v step', round(d['ms_per_denoise_step'],3), round(d['value'],4))"
done | tee $O/ab.txt
