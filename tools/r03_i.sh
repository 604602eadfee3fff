#!/bin/bash
# round 3, call I: wave-specialised GEGLU kernel -- parity, per-shape A/B, whole-step A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -m gpu -q -p no:cacheprovider -x -k "geglu" > $O/t.log 2>&1; echo "tests rc=$?"; tail -5 $O/t.log
export PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so
for ws in 0 1; do PP_GEGLU_WS=$ws timeout 300 python tools/geglu_ws_ab.py 2>&1 | grep WS= ; done | tee $O/ab.txt
for rep in 1 2; do for ws in 0 1; do
PP_GEGLU_WS=$ws timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WS=$ws rep$rep ms/step', round(d['ms_per_denoise_step'],4), 'img/s', round(d['value'],3))"
done; done | tee -a $O/ab.txt
