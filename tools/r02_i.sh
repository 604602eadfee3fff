#!/bin/bash
# attention: QK^T MFMA chains of the two 32-key blocks interleaved (PP_ATTN_ILV) against block order.  The switch lived
# only in the experiment build: no effect, reverted (profiles/r02_rejected_experiments.txt, item 11).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02i
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -p no:cacheprovider -k "attention" > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
for i in 1 2 3; do
  for v in 0 1; do echo -n "ILV=$v "; PP_ATTN_ILV=$v timeout 120 python tools/attn_ablate.py one; done
done 2>&1 | tee $O/ab.txt
for i in 1 2; do for v in 0 1; do
  PP_ATTN_ILV=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ILV=$v step', round(d['ms_per_denoise_step'],3))"
done; done | tee -a $O/ab.txt
