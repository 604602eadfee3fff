#!/bin/bash
# Evidence run on the GPU box (one gpurun call:  gpurun --timeout 2400 -- bash tools/evidence.sh [tests|profiles]): GPU test suite (arg "tests": + the slow config-parity file), smoke, rocprofv3
# kernel trace + HBM-traffic counters of the headline command, headline bench with cpu_baseline + live roofline,
# configs 3 / 4 / 5 and fp16.  -> gpurun_out/$ROUND/ (default r04); the profiles the judge reads are copied to profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/parity_achieved.txt
if [ "${1:-}" = "profiles" ]; then
  echo "(profiles only: test suite and smoke skipped)"
elif [ "${1:-}" = "parity" ]; then      # only the slow config-parity file (the rest of the suite ran in another session on this build)
  timeout 1200 python -m pytest tests/test_config_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
elif [ "${1:-}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
else
  timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --ignore=tests/test_config_parity_gpu.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
fi
[ "${1:-}" = "profiles" ] || { timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log; }
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/kernel_stats.txt "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline" > /dev/null 2>&1
python tools/step_timeline.py $DB $O/step_timeline.txt > /dev/null 2>&1; rm -rf $O/prof; head -14 $O/kernel_stats.txt; head -3 $O/step_timeline.txt
bash tools/hbm_traffic.sh > $O/hbm.log 2>&1; cp gpurun_out/hbm_traffic.json $O/hbm_traffic.json
cp $O/kernel_stats.txt profiles/${R}_kernel_stats.txt; cp $O/hbm_traffic.json profiles/${R}_hbm_traffic.json     # (bench.py reads these)
timeout 400 python bench.py --dump-launches $O/launches.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1800 $O/bench.json
timeout 600 python bench.py --config v2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_v2.json 2>> $O/bench.err; echo "v2 rc=$?"
timeout 600 python bench.py --config controlnet --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_controlnet.json 2>> $O/bench.err; echo "cn rc=$?"
timeout 900 python bench.py --config v2 --latent 128 --per-gpu 2 --denoise-steps 30 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err; echo "cfg5 rc=$?"
timeout 600 python bench.py --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_fp16.json 2>> $O/bench.err; echo "fp16 rc=$?"
for f in bench bench_v2 bench_controlnet bench_config5 bench_fp16; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value'],4), 'img/s', round(d['ms_per_denoise_step'],3), 'ms/step util', round(d['unet_step_mfma_util'],4), 'frac', d.get('roofline',{}).get('frac'))
except Exception as e: print('$f', 'ERR', e)
PY
done
tail -5 $O/bench.err
if [ "${PMC:-0}" = "1" ]; then
  # counters of the shipping fused conv (first shape of tools/conv_gn_pmc.sh: 64x64, 320 -> 320) and of the fused feed-forward,
  # ablations of the feed-forward's lab build (PP_FF_DBG: 1 no weight DMA, 8 no GEGLU, 13 MFMAs only, 16 no barriers)
  SHAPES_ONE=1 bash tools/conv_gn_pmc.sh > $O/conv_gn_pmc.log 2>&1; cp gpurun_out/conv_gn_pmc.txt $O/gemm_pmc.txt
  ONLY=fused bash tools/ff_pmc.sh > $O/ff_pmc.log 2>&1; cp gpurun_out/ff_pmc.txt $O/ff_fused_pmc.txt
  { echo "# tools/ff_one.py 32768 on libpp_hip_lab.so (same sources as $(python -c 'from powerpaint_amd import _lib; print(_lib.build_id())')), PP_FF_DBG ablations of ff_fused8_kernel; eager launches (~8 us of launch gap included)";
    for d in 0 1 8 9 13 16; do echo "## PP_FF_DBG=$d"; PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so PP_FF_DBG=$d timeout 100 python tools/ff_one.py 32768 2>&1 | grep -E "fused:|chain:|fused4:" | tail -3; done; } > $O/ff_fused.txt 2>&1
fi
cat gpurun_out/parity_achieved.txt 2>/dev/null | grep -i "free-running\|golden\|loop\|smoke" | head -40
