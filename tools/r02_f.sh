#!/bin/bash
# conv_out MFMA kernel (32 pixels / workgroup, hand-pipelined taps): parity, then the evidence set again
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02f
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_vae.py -q -p no:cacheprovider -m gpu -k "smallcout or vae or decode or encode" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/t_ops.log
timeout 600 python -m pytest tests/test_real_shapes_gpu.py -q -x -p no:cacheprovider -k "config2" > $O/t_real.log 2>&1; echo "real rc=$?"; tail -2 $O/t_real.log
bash tools/r02_evidence.sh
