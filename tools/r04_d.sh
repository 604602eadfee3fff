#!/bin/bash
# Round-4 session D: the hand-scheduled in-burst normalisation of conv_gn.hip.  Op parity, per-shape timing, interleaved
# headline benches fused / unfused.  -> gpurun_out/r04d/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_ship.log 2>&1; rc=$?; echo "op tests (shipping) rc=$rc"; grep -E "passed|failed" $O/op_ship.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op_ship.log | head -30
if [ $rc -eq 124 ]; then echo "HANG"; exit 0; fi
timeout 300 python tools/conv_gn_shapes.py --out $O/shapes_ship.json > $O/shapes_ship.txt 2>&1; echo "ship rc=$?"; grep -v amdgpu.ids $O/shapes_ship.txt
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  timeout 300 $B > $O/bench_fused_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_FUSE_GN_CONV=0 timeout 300 $B > $O/bench_unfused_$i.json 2>> $O/bench.err
done
for f in fused_1 unfused_1 fused_2 unfused_2; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value'],4), 'img/s', round(d['ms_per_denoise_step'],3), 'ms/step')
except Exception as e: print('$f', 'ERR', e)
PY
done
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-launches $O/launches_fused.json > $O/bench_roofline.json 2>> $O/bench.err
tail -3 $O/bench.err
exit 0
