#!/bin/bash
# round 2, run A: ping-pong GEMM tiles -- correctness of the op tests, then the A/B over the UNet's launch shapes
mkdir -p gpurun_out/r02a
python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or conv" > gpurun_out/r02a/pytest_ops.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest_ops.log
tail -5 gpurun_out/r02a/pytest_ops.log
python tools/gemm_pp_bench.py --tiles 33,53,31,54,44,24 --splitk 1 --min-gflop 10 --rounds 2 --json gpurun_out/r02a/pp_bench.json > gpurun_out/r02a/pp_bench.log 2>&1
tail -150 gpurun_out/r02a/pp_bench.log
