timeout 900 python -m pytest tests/test_fused_combine_gpu.py tests/test_golden.py tests/test_abi.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -5 > gpurun_out/r7e_test.txt
tail -5 gpurun_out/r7e_test.txt
L="PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so"
bash tools/step_ab.sh r7e 4 "ship|" "sep|PP_LAB=1 PP_FUSED_COMBINE=0"
