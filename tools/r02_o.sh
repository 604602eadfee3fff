#!/bin/bash
# cross-attention: K / V reuse over two 128-query blocks per workgroup (PP_ATTN_QREP) -- parity forced on, then A/B
set -u
cd "$(dirname "$0")/.."
PP_ATTN_QREP=2 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -p no:cacheprovider -k "attention" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "attention" 2>&1 | tail -1
for i in 1 2; do for v in 1 0; do
  PP_ATTN_QREP=$v timeout 300 python bench.py --steps 3 --no-cpu-baseline --dump-launches /tmp/l$v.json 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=json.load(open('/tmp/l$v.json')); a=[round(e['ms']*1000,1) for e in L if e['what']=='attention']
print('QREP=$v step', round(d['ms_per_denoise_step'],3), 'cross-attn 64x64:', a[1], a[3], ' 32x32:', a[5])"
done; done
