#!/bin/bash
# Round-2 evidence run on the GPU box (no test suite: that is its own call): headline bench with cpu_baseline + roofline,
# configs 3 / 4 / 5, rocprofv3 kernel trace + HBM-traffic counters of the headline command.  -> gpurun_out/r02/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
if [ "${1:-}" = "tests" ]; then
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
exit 0
fi
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline" > /dev/null 2>&1; rm -rf $O/prof; head -12 $O/kernel_stats.txt
cp $O/kernel_stats.txt profiles/r02_kernel_stats.txt      # bench.py reads the kernel-only time base from here
bash tools/hbm_traffic.sh > $O/hbm.log 2>&1; cp gpurun_out/hbm_traffic.json $O/hbm_traffic.json; cp gpurun_out/hbm_traffic.json profiles/r02_hbm_traffic.json
timeout 400 python bench.py --dump-launches $O/launches.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
timeout 600 python bench.py --config v2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_v2.json 2>> $O/bench.err; echo "v2 rc=$?"
timeout 600 python bench.py --config controlnet --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_controlnet.json 2>> $O/bench.err; echo "cn rc=$?"
timeout 900 python bench.py --config v2 --latent 128 --per-gpu 2 --denoise-steps 30 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err; echo "cfg5 rc=$?"; tail -c 700 $O/bench_config5.json
timeout 600 python bench.py --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_fp16.json 2>> $O/bench.err; echo "fp16 rc=$?"; tail -c 400 $O/bench_fp16.json
tail -5 $O/bench.err
