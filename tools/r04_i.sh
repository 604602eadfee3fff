#!/bin/bash
# Round-4 session I: GroupNorm apply inside the split-K combine (PPGemmArgs.gn_next_*).  Op parity (bit-exact), network
# parity through it, interleaved headline benches.  -> gpurun_out/r04i/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04i
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=120 -k "combine" > $O/op.log 2>&1; rc=$?; echo "op tests rc=$rc"; grep -E "passed|failed" $O/op.log | tail -2; grep -E "^(FAILED|ERROR)|Error" $O/op.log | head -12 | cut -c1-300
if [ $rc -eq 124 ]; then echo HANG; exit 0; fi
timeout 1200 python -m pytest tests/test_real_shapes_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $O/models.log 2>&1; echo "models rc=$?"; grep -E "passed|failed" $O/models.log | tail -2; grep -E "^(FAILED|ERROR)" $O/models.log | head -10 | cut -c1-300
grep "headline parity" gpurun_out/parity_r04.txt | grep -v "step [0-9]" | tail -3 | cut -c1-250
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_new_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_GN_NEXT=0 timeout 300 $B > $O/bench_old_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for m in ("new", "old"):
    r = []
    for i in (1, 2, 3):
        try:
            d = json.loads(open('$O/bench_%s_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append('%.3f ms (%s launches)' % (d['ms_per_denoise_step'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append('ERR ' + str(e)[:60])
    print(m, '  '.join(r))
PY
tail -3 $O/bench.err
exit 0
