#!/bin/bash
# One box, interleaved: the shipping plan, each round-4 fusion switched off alone, and all of them off.  -> gpurun_out/r04decomp/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04decomp
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 $B > /dev/null 2>&1      # warm the box
for i in 1 2; do
  timeout 300 $B > $O/b_ship_$i.json 2>> $O/bench.err
  env $LAB timeout 300 $B > $O/b_lab_$i.json 2>> $O/bench.err
  env $LAB PP_FUSE_GN_CONV=0 timeout 300 $B > $O/b_noconvgn_$i.json 2>> $O/bench.err
  env $LAB PP_TFRONT=0 timeout 300 $B > $O/b_notfront_$i.json 2>> $O/bench.err
  env $LAB PP_GN_NEXT=0 timeout 300 $B > $O/b_nognnext_$i.json 2>> $O/bench.err
  env $LAB PP_XATTN_WIDE=0 timeout 300 $B > $O/b_nowide_$i.json 2>> $O/bench.err
  env $LAB PP_CONV_GN_SK2=0 timeout 300 $B > $O/b_nosk2_$i.json 2>> $O/bench.err
  env $LAB PP_FUSE_GN_CONV=0 PP_TFRONT=0 PP_GN_NEXT=0 PP_XATTN_WIDE=0 timeout 300 $B > $O/b_r3plan_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for m in ("ship", "lab", "noconvgn", "notfront", "nognnext", "nowide", "nosk2", "r3plan"):
    r = []
    for i in (1, 2):
        try:
            d = json.loads(open('$O/b_%s_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append('%.3f ms (%s)' % (d['ms_per_denoise_step'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append('ERR ' + str(e)[:60])
    print('%-9s' % m, '  '.join(r))
PY
tail -2 $O/bench.err
exit 0
