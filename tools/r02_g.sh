#!/bin/bash
# does HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) shorten the per-kernel fixed time inside graph replays?
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02g
mkdir -p $O
for i in 1 2; do
  for v in 0 1; do
    HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEV_KERNARG=$v', round(d['ms_per_denoise_step'],3), round(d['value'],3))"
  done
done | tee $O/ab.txt
