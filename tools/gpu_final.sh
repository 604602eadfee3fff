#!/bin/bash
# Round-end evidence run on the GPU box: full GPU test suite, smoke, headline bench (with cpu_baseline + roofline),
# the other two configs, a rocprofv3 kernel trace of the headline command.  Everything lands in gpurun_out/final/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py --dump-launches $O/launches.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
timeout 600 python bench.py --config v2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_v2.json 2>> $O/bench.err; echo "v2 rc=$?"
timeout 600 python bench.py --config controlnet --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_controlnet.json 2>> $O/bench.err; echo "cn rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt "bench.py --steps 1 --warmup 1" > /dev/null 2>&1; rm -rf $O/prof; head -25 $O/kernel_stats.txt
timeout 300 python tools/plan_ablate.py > $O/plan_ablate.txt 2>&1; tail -14 $O/plan_ablate.txt
timeout 200 python tools/vae_time.py > $O/vae_time.txt 2>&1; tail -5 $O/vae_time.txt
