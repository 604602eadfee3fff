#!/bin/bash
# Round-4 session J: 256-row tiles x 2 K splits for the long-K fused convs of the 32x32 level.  -> gpurun_out/r04j/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04j
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op.log 2>&1; echo "op tests rc=$?"; grep -E "passed|failed" $O/op.log | tail -1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  for m in 12 0 9 16; do
    env $LAB PP_CONV_GN_SK2=$m timeout 300 $B > $O/bench_m${m}_$i.json 2>> $O/bench.err
  done
done
python - <<PY
import json
for m in (12, 0, 9, 16):
    r = []
    for i in (1, 2, 3):
        try:
            d = json.loads(open('$O/bench_m%d_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append('%.3f ms (%s)' % (d['ms_per_denoise_step'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append('ERR ' + str(e)[:60])
    print('sk2 from %2d chunks:' % m, '  '.join(r))
PY
tail -3 $O/bench.err
exit 0
