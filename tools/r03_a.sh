#!/bin/bash
# round 3, call A: the new parity tests + live-roofline bench line
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_r03.txt
timeout 900 python -m pytest tests/test_config_parity_gpu.py -q -s -p no:cacheprovider --timeout=600 > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
timeout 600 python -m pytest tests/test_clip.py tests/test_golden.py tests/test_host.py -m gpu -q -p no:cacheprovider --timeout=600 > $O/t2.log 2>&1; echo "clip/golden rc=$?"; tail -5 $O/t2.log
timeout 400 python bench.py --dump-launches $O/launches.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json; tail -5 $O/bench.err
cat gpurun_out/parity_r03.txt
