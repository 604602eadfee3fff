L="PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so"
bash tools/step_ab.sh r7n 4 "ship|" "r5plan|$L PP_CONV_GN_ROUTE=15 PP_CONV_RAW=0 PP_FUSED_COMBINE=0"
BENCH_ARGS="--config v2 --steps 2" bash tools/step_ab.sh r7n_v2 2 "ship|" "r5plan|$L PP_CONV_GN_ROUTE=15 PP_CONV_RAW=0 PP_FUSED_COMBINE=0"
BENCH_ARGS="--config controlnet --steps 2" bash tools/step_ab.sh r7n_cn 2 "ship|" "r5plan|$L PP_CONV_GN_ROUTE=15 PP_CONV_RAW=0 PP_FUSED_COMBINE=0"
