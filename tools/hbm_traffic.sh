#!/bin/bash
# HBM traffic of the headline command from the L2's memory-side counters, one --pmc pass per counter as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Output: gpurun_out/hbm/{FETCH_SIZE,WRITE_SIZE}/...db -> tools/hbm_traffic.py -> gpurun_out/hbm_traffic.json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/hbm
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $O/$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph > $O/$c.log 2>&1
  echo "$c rc=$?"
done
cd /root/repo && cd /root/repo && python tools/hbm_traffic.py $O gpurun_out/hbm_traffic.json && rm -rf $O/FETCH_SIZE $O/WRITE_SIZE   # (the counter databases are ~25 MB each: keep the merge-back under gpurun's 64 MiB)
