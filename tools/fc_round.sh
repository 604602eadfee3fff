timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --ignore=tests/test_config_parity_gpu.py -x 2>&1 | tail -8 > gpurun_out/r6z_test.txt
tail -8 gpurun_out/r6z_test.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_denoise_step'], d['launches_per_denoise_step'], d['roofline']['frac'], d['unet_step_mfma_util'])"
