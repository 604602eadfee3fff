#!/bin/bash
# Round 4 against the round-3 launch plan on ONE box, interleaved: every round-4 fusion off (lab switches) vs the shipping
# plan.  -> gpurun_out/r04total/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04total
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_r4_$i.json 2>> $O/bench.err
  env $LAB PP_FUSE_GN_CONV=0 PP_TFRONT=0 PP_GN_NEXT=0 PP_XATTN_WIDE=0 timeout 300 $B > $O/bench_r3plan_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for m in ("r4", "r3plan"):
    r = []
    for i in (1, 2, 3):
        try:
            d = json.loads(open('$O/bench_%s_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append('%.3f ms / %.3f img/s (%s launches)' % (d['ms_per_denoise_step'], d['value'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append('ERR ' + str(e)[:60])
    print(m, '  '.join(r))
PY
tail -2 $O/bench.err
exit 0
