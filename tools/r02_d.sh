#!/bin/bash
# conv_in on the implicit-GEMM kernel + GELU sign trick + half key tail of the cross-attention: parity, then the step
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02d
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "attention or geglu or gelu" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/t_ops.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_real_shapes_gpu.py -q -x -p no:cacheprovider -s > $O/t_models.log 2>&1; echo "models rc=$?"; tail -3 $O/t_models.log; grep "real-shape parity" $O/t_models.log
timeout 400 python bench.py --no-cpu-baseline --dump-launches $O/launches.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1200 $O/bench.json; tail -3 $O/bench.err
