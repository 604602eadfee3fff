#!/bin/bash
# round 3, call J: ppt-v1 / ControlNet pipelines with a 4-channel UNet (pp_latent_blend) + the golden suite
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_golden.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "golden or latent_blend or ddim or sched or 4_channel or reference_call" > $O/t.log 2>&1; echo "tests rc=$?"; tail -15 $O/t.log
