mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "minimal_filtering or gru_convolutions or slice_reg or fused_level_one" > gpurun_out/rpr_tests.log 2>&1; tail -2 gpurun_out/rpr_tests.log
timeout 200 tools/step_prof.sh rpr1 2>&1 | grep -E "wino|conv1|tail|ms" | head -12
ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.rpr0.so timeout 200 tools/step_prof.sh rpr0 2>&1 | grep -E "wino|conv1|ms" | head -12
