#!/bin/bash
# Round-4 session G: the wide fused cross-attention block (C = 640 / 1280).  Op parity, then interleaved headline benches
# with / without it.  -> gpurun_out/r04g/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider --timeout=120 -k "fused_cross_attention_block and 640-True-dtype0" > $O/op_first.log 2>&1; rc=$?; echo "first cases rc=$rc"; tail -15 $O/op_first.log | cut -c1-300
if [ $rc -eq 124 ]; then echo "HANG"; exit 0; fi
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider --timeout=120 -k "fused_cross_attention_block" > $O/op.log 2>&1; echo "op tests rc=$?"; grep -E "passed|failed" $O/op.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op.log | head -30 | cut -c1-250
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 300 $B > $O/bench_wide_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_XATTN_WIDE=0 timeout 300 $B > $O/bench_chain_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_XATTN_WIDE_C=1280 timeout 300 $B > $O/bench_w1280_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for m in ("wide", "chain", "w1280"):
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
python tools/step_timeline.py $DB $O/step_timeline.txt > $O/timeline.log 2>&1; echo "timeline rc=$?"; grep -E "xattn|attn_fwd" $O/step_timeline.txt | head -12
rm -rf $O/prof
tail -3 $O/bench.err
exit 0
