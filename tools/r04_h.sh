#!/bin/bash
# Round-4 session H: the pipelined attention loop at d = 80, the multi-block time-embedding row copy, the scsh store fix.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04h
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider --timeout=120 -k "test_attention or embed or splice" > $O/op.log 2>&1; echo "attention op tests rc=$?"; grep -E "passed|failed" $O/op.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op.log | head -20 | cut -c1-250
timeout 300 python -m pytest tests/test_conv_gn_gpu.py tests/test_task_tokens.py -q -p no:cacheprovider --timeout=120 -m gpu > $O/op2.log 2>&1; echo "conv_gn + token tests rc=$?"; grep -E "passed|failed" $O/op2.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op2.log | head -20 | cut -c1-250
timeout 600 python -m pytest tests/test_headline_parity_gpu.py -q -p no:cacheprovider --timeout=500 > $O/headline.log 2>&1; echo "headline (fixture) rc=$?"; grep -E "passed|failed" $O/headline.log | tail -2; grep "headline parity" $O/headline.log | grep -v "step [0-9]" | cut -c1-260
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_new_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so PP_ATTN_PIPE80=0 timeout 300 $B > $O/bench_phased80_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for m in ("new", "phased80"):
    r = []
    for i in (1, 2, 3):
        try:
            d = json.loads(open('$O/bench_%s_%d.json' % (m, i)).read().strip().splitlines()[-1])
            r.append('%.3f ms (%s launches)' % (d['ms_per_denoise_step'], d.get('launches_per_denoise_step')))
        except Exception as e:
            r.append('ERR ' + str(e)[:60])
    print(m, '  '.join(r))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
DB=$(find $O/prof -name "*.db" | head -1)
python tools/step_timeline.py $DB $O/step_timeline.txt > $O/timeline.log 2>&1; echo "timeline rc=$?"; grep -E "attn_|embed_splice|sum of" $O/step_timeline.txt | head -12
rm -rf $O/prof
tail -3 $O/bench.err
exit 0
