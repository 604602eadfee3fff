timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 2>&1 | tail -8 > gpurun_out/r7b_test.txt
tail -8 gpurun_out/r7b_test.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
L="PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so"
bash tools/step_ab.sh r7b 3 "ship|" "r5plan|$L PP_CONV_GN_ROUTE=15 PP_CONV_RAW=0 PP_FUSED_COMBINE=0"
