#!/bin/bash
# Round-4 session E: which convs to fuse -- interleaved headline benches over the routing classes of pp_conv_gn_preferred
# (lab build, PP_CONV_GN_ROUTE).  -> gpurun_out/r04e/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  for m in 0 1 5 7 3 15 31; do
    env $LAB PP_CONV_GN_ROUTE=$m timeout 300 $B > $O/bench_m${m}_$i.json 2>> $O/bench.err
  done
done
python - <<PY
import json
for m in (0, 1, 5, 7, 3, 15, 31):
    r = []
    for i in (1, 2, 3):
        try:
            d = json.loads(open('$O/bench_m%d_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append((d['ms_per_denoise_step'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append(('ERR', str(e)[:40]))
    print('route mask %2d:' % m, '  '.join('%s ms (%s launches)' % (('%.3f' % a) if not isinstance(a, str) else a, b) for a, b in r))
PY
tail -3 $O/bench.err
exit 0
