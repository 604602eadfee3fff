#!/bin/bash
# Per-dispatch timelines of the step with and without the wide fused cross-attention block, same box.  -> gpurun_out/r04wide/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04wide
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in wide chain; do
  if [ $v = chain ]; then export PP_LAB=1 PP_XATTN_WIDE=0; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$v -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof_$v.log 2>&1)
  DB=$(find $O/prof_$v -name "*.db" | head -1)
  python tools/step_timeline.py $DB $O/timeline_$v.txt > /dev/null 2>&1
  rm -rf $O/prof_$v
  head -2 $O/timeline_$v.txt
done
exit 0
