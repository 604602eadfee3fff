#!/bin/bash
# Round-4 first GPU session: the fused norm -> SiLU -> conv kernel (conv_gn.hip).  Op parity (ping-pong and lock-step
# loops), network parity through it, interleaved same-box A/B of the headline bench with the fusion on / off, per-launch
# timings, kernel trace.  Everything lands in gpurun_out/r04a/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
python -c "import torch;print('torch',torch.__version__,'gpu',torch.cuda.get_device_name(0))" 2>&1 | tail -1
# 1. one small case first, short leash: a hang here must not eat the session
timeout 240 python -m pytest tests/test_conv_gn_gpu.py -q -x -p no:cacheprovider -k "levels and 64-64-64-320" > $O/op_first.log 2>&1; rc=$?; echo "first case rc=$rc"; tail -5 $O/op_first.log
if [ $rc -eq 124 ]; then echo "HANG in the fused kernel: running the rest with the fusion off"; export PP_LAB=1 PP_FUSE_GN_CONV=0; fi
if [ $rc -ne 124 ]; then
  timeout 900 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_pp.log 2>&1; echo "op tests (ping-pong) rc=$?"; grep -E "passed|failed" $O/op_pp.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op_pp.log | head -40
  env $LAB PP_CONV_GN_PP=0 timeout 900 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_lock.log 2>&1; echo "op tests (lock-step) rc=$?"; grep -E "passed|failed" $O/op_lock.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op_lock.log | head -20
fi
# 2. the networks through it
timeout 1500 python -m pytest tests/test_real_shapes_gpu.py tests/test_golden.py -m gpu -q -p no:cacheprovider --timeout=900 > $O/models.log 2>&1; echo "models rc=$?"; grep -E "passed|failed" $O/models.log | tail -2; grep -E "^(FAILED|ERROR)" $O/models.log | head -20
# 3. interleaved A/B of the headline bench: fused / unfused / fused / unfused
for i in 1 2; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_fused_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_FUSE_GN_CONV=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_unfused_$i.json 2>> $O/bench.err
done
env $LAB PP_CONV_GN_PP=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_fused_lockstep.json 2>> $O/bench.err
for f in fused_1 unfused_1 fused_2 unfused_2 fused_lockstep; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value'],4), 'img/s', round(d['ms_per_denoise_step'],3), 'ms/step util', round(d['unet_step_mfma_util'],4))
except Exception as e: print('$f', 'ERR', e)
PY
done
# 4. per-launch timings (eager, event pairs) fused and unfused, live roofline
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-launches $O/launches_fused.json > $O/bench_roofline.json 2>> $O/bench.err; tail -c 1500 $O/bench_roofline.json
PP_LAB=1 PP_FUSE_GN_CONV=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --dump-launches $O/launches_unfused.json > /dev/null 2>> $O/bench.err
# 5. kernel trace of the fused step
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?")
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline" > /dev/null 2>&1; rm -rf $O/prof; head -24 $O/kernel_stats.txt
tail -5 $O/bench.err
exit 0
