#!/bin/bash
# Round-4 session C: per-shape timing of the fused norm -> SiLU -> conv launch (halo strips per wave now a template
# parameter) in its variants, then the headline parity file.  -> gpurun_out/r04c/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LAB="PP_LAB=1 PP_LIB=$PWD/powerpaint_amd/libpp_hip_lab.so"
timeout 600 python -m pytest tests/test_conv_gn_gpu.py -q -p no:cacheprovider --timeout=300 > $O/op_ship.log 2>&1; rc=$?; echo "op tests (shipping) rc=$rc"; grep -E "passed|failed" $O/op_ship.log | tail -2; grep -E "^(FAILED|ERROR)" $O/op_ship.log | head -30
if [ $rc -eq 124 ]; then echo "HANG"; exit 0; fi
timeout 300 python tools/conv_gn_shapes.py --out $O/shapes_ship.json > $O/shapes_ship.txt 2>&1; echo "ship rc=$?"; cat $O/shapes_ship.txt | grep -v amdgpu.ids
for nm in 0 2 3; do
  env $LAB PP_CONV_GN_NMODE=$nm timeout 300 python tools/conv_gn_shapes.py --out $O/shapes_nm$nm.json > $O/shapes_nm$nm.txt 2>&1; echo "nmode $nm rc=$?"
  grep -E "fused|per UNet" $O/shapes_nm$nm.txt | awk '{print}' | cut -c1-70
done
timeout 1100 python -m pytest tests/test_headline_parity_gpu.py -q -x -p no:cacheprovider --timeout=1000 > $O/headline.log 2>&1; echo "headline parity rc=$?"; tail -3 $O/headline.log
cp gpurun_out/parity_r04.txt $O/ 2>/dev/null; cat gpurun_out/parity_r04.txt 2>/dev/null | tail -20
exit 0
