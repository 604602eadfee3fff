#!/bin/bash
mkdir -p gpurun_out/r02c
python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or conv" > gpurun_out/r02c/pytest_ops.log 2>&1
tail -3 gpurun_out/r02c/pytest_ops.log
python tools/gemm_pp_bench.py --tiles 33,53,31,54 --splitk 1 --min-gflop 30 --min-m 8192 --rounds 3 --json gpurun_out/r02c/pp_bench.json > gpurun_out/r02c/pp_bench.log 2>&1
grep -v "^/opt" gpurun_out/r02c/pp_bench.log
