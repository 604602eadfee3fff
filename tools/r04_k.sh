#!/bin/bash
# Round-4 session K: the fused transformer front end (csrc/tfront.hip).  -> gpurun_out/r04k/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04k
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider --timeout=120 -k "front_end and 3-128" > $O/op_first.log 2>&1; rc=$?; echo "first rc=$rc"; tail -12 $O/op_first.log | cut -c1-300
if [ $rc -eq 124 ]; then echo HANG; exit 0; fi
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider --timeout=120 -k "front_end" > $O/op.log 2>&1; echo "op tests rc=$?"; grep -E "passed|failed" $O/op.log | tail -1; grep -E "^(FAILED|ERROR)|AssertionError" $O/op.log | head -12 | cut -c1-300
timeout 900 python -m pytest tests/test_real_shapes_gpu.py tests/test_headline_parity_gpu.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $O/models.log 2>&1; echo "models rc=$?"; grep -E "passed|failed" $O/models.log | tail -1; grep -E "^(FAILED|ERROR)" $O/models.log | head -8 | cut -c1-300
grep "headline parity" gpurun_out/parity_r04.txt | grep -v "step [0-9]" | tail -3 | cut -c1-220
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_new_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_TFRONT=0 timeout 300 $B > $O/bench_old_$i.json 2>> $O/bench.err
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
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
python tools/step_timeline.py $DB $O/step_timeline.txt > $O/timeline.log 2>&1; grep -E "tfront|sum of" $O/step_timeline.txt | head -4
rm -rf $O/prof
tail -3 $O/bench.err
exit 0
