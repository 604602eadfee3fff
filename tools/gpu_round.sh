#!/bin/bash
# One GPU-box session: op parity, model parity, bench, (optional) rocprof.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [stages...]   stages: ops models bench prof
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${@:-ops models bench}"
python -c "import torch;print('torch',torch.__version__,'gpu',torch.cuda.get_device_name(0))" 2>&1 | tee gpurun_out/env.log
nproc | tee -a gpurun_out/env.log
for s in $STAGES; do
  case $s in
    ops)    timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -rA --timeout=300 -p no:cacheprovider > gpurun_out/pytest_ops.log 2>&1; echo "ops rc=$?";;
    models) timeout 1500 python -m pytest tests/test_models_gpu.py -m gpu -q -rA --timeout=900 -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "models rc=$?";;
    bench)  timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.log;;
    benchq) timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.log;;
    prof)   cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -o r1 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof.log 2>&1; echo "prof rc=$?"; cd /root/repo;;
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log;;
  esac
done
for f in gpurun_out/pytest_ops.log gpurun_out/pytest_models.log; do [ -f $f ] && { echo "== $f"; grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $f | tail -80; }; done
[ -s gpurun_out/bench.err ] && { echo "== bench.err"; tail -30 gpurun_out/bench.err; }
exit 0
