#!/bin/bash
# Round-4 session F: per-dispatch timeline of one denoise step (kernel durations and inter-kernel gaps).  -> gpurun_out/r04f/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
DB=$(find $O/prof -name "*.db" | head -1)
python tools/step_timeline.py $DB $O/step_timeline.txt > $O/timeline.log 2>&1; echo "timeline rc=$?"; head -60 $O/step_timeline.txt
python tools/prof_summary.py $DB $O/kernel_stats.txt "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline" > /dev/null 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$DB")
print([r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")][:60])
PY
cp $DB $O/results.db 2>/dev/null; rm -rf $O/prof
tail -3 $O/timeline.log
exit 0
