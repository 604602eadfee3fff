#!/bin/bash
# round 3, call H: the shipping attention default on every dtype / shape + config 2 / config 5 bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_config_parity_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -k "attention or callback or unet_tiny or fp16" > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
timeout 600 python bench.py --config v2 --latent 128 --per-gpu 2 --denoise-steps 30 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],4))"
