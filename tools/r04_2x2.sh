#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r042x2
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 $B > /dev/null 2>&1
for i in 1 2 3; do
  for t in 1 0; do for w in 1 0; do
    env $LAB PP_TFRONT=$t PP_XATTN_WIDE=$w timeout 300 $B > $O/b_t${t}w${w}_$i.json 2>> $O/bench.err
  done; done
done
python - <<PY
import json
for t in (1, 0):
    for w in (1, 0):
        r = []
        for i in (1, 2, 3):
            d = json.loads(open('$O/b_t%dw%d_%d.json' % (t, w, i)).read().strip().splitlines()[-1])
            r.append('%.3f' % d['ms_per_denoise_step'])
        print('tfront %d wide %d:' % (t, w), ' '.join(r), '(%s launches)' % d.get('launches_per_denoise_step'))
PY
python -c "
import torch;print(torch.cuda.get_device_name(0))"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
exit 0
