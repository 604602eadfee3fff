timeout 900 python -m pytest tests/test_conv_gn_gpu.py tests/test_golden.py tests/test_real_shapes_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -5 > gpurun_out/r7g_test.txt
tail -5 gpurun_out/r7g_test.txt
L="PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so"
bash tools/step_ab.sh r7g 4 "ship|" "prev|PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_prev.so"
