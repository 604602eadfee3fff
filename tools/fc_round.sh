L="PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so"
bash tools/step_ab.sh r7f 4 "ship|" "lab4|$L" "fused8|$L PP_FUSED_COMBINE_SPLITS=8" "fused2|$L PP_FUSED_COMBINE_SPLITS=2"
