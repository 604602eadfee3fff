#!/bin/bash
# Round-4 A/B session for the fused norm -> SiLU -> conv kernel: op parity of the shipping variant (ping-pong, normalisation
# inside the MFMA burst) and of the lab variants, then interleaved headline benches.  -> gpurun_out/r04b/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_ship.log 2>&1; rc=$?; echo "op tests (shipping) rc=$rc"; grep -E "passed|failed" $O/op_ship.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op_ship.log | head -30
if [ $rc -eq 124 ]; then echo "HANG"; exit 0; fi
env $LAB PP_CONV_GN_NMODE=0 timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_nm0.log 2>&1; echo "op tests (NMODE 0) rc=$?"; grep -E "passed|failed" $O/op_nm0.log | tail -1
env $LAB PP_CONV_GN_PP=0 timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_lock.log 2>&1; echo "op tests (lock-step) rc=$?"; grep -E "passed|failed" $O/op_lock.log | tail -1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  timeout 300 $B > $O/bench_fused_$i.json 2>> $O/bench.err
  PP_LAB=1 PP_FUSE_GN_CONV=0 timeout 300 $B > $O/bench_unfused_$i.json 2>> $O/bench.err
  env $LAB PP_CONV_GN_NMODE=0 timeout 300 $B > $O/bench_nm0_$i.json 2>> $O/bench.err
  env $LAB PP_CONV_GN_PP=0 timeout 300 $B > $O/bench_lock_$i.json 2>> $O/bench.err
done
for f in fused_1 unfused_1 nm0_1 lock_1 fused_2 unfused_2 nm0_2 lock_2; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value'],4), 'img/s', round(d['ms_per_denoise_step'],3), 'ms/step')
except Exception as e: print('$f', 'ERR', e)
PY
done
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-launches $O/launches_fused.json > $O/bench_roofline.json 2>> $O/bench.err
PP_LAB=1 PP_FUSE_GN_CONV=0 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-launches $O/launches_unfused.json > /dev/null 2>> $O/bench.err
python - <<PY
import json, collections
def load(f):
    L=json.load(open(f)); agg=collections.OrderedDict()
    for l in L:
        a=agg.setdefault(l['what'].replace(' cat',''),[0,0.0]); a[0]+=1; a[1]+=l['ms']*1000
    return agg, sum(l['ms'] for l in L)*1000, len(L)
F,tf,nf=load('$O/launches_fused.json'); U,tu,nu=load('$O/launches_unfused.json')
print('eager sum us: fused',round(tf),nf,'launches; unfused',round(tu),nu)
tot=0
for k,(n,t) in F.items():
    if k.startswith('conv3x3'):
        u=U.get(k,[1,0]); d=t/n-u[1]/u[0]; tot+=d*n
        print('%-48s n=%d fused %7.1f unfused %7.1f delta %6.1f'%(k,n,t/n,u[1]/u[0],d))
print('conv delta sum',round(tot,1),'us; gn apply unfused',U['groupnorm_apply'],'fused',F['groupnorm_apply'])
PY
tail -3 $O/bench.err
exit 0
