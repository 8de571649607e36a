#!/bin/bash
cd "$(dirname "$0")/.."
g=gpurun_out/r06m
mv ada-mvs_amd/libadamvs_hip.bxc_hwregs.so /tmp/hw.so
python tools/experiments/bx3_costreg_timing/time_variants.py > ${g}_bxc.txt 2>&1; cat ${g}_bxc.txt
ADAMVS_LIB_PATH=/tmp/hw.so python tools/experiments/bx3_costreg_timing/hwregs.py 2>&1 | tail -4
