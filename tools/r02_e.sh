#!/bin/bash
# strength < 1 / guess_mode / eta / begin index: the new parity tests, then the rest of the pipeline-level suites
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02e
mkdir -p $O
timeout 900 python -m pytest tests/test_golden.py tests/test_ops_gpu.py -q -p no:cacheprovider -m gpu -k "golden or strength or guess or eta or sched or late or pndm or unipc or smallcout or reproduces" > $O/t_new.log 2>&1; echo "new rc=$?"; tail -15 $O/t_new.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_controller.py tests/test_fp16_gpu.py -q -x -m gpu -p no:cacheprovider > $O/t_models.log 2>&1; echo "models rc=$?"; tail -4 $O/t_models.log
