#!/bin/bash
cd "$(dirname "$0")/.."
g=gpurun_out/r06h
python tools/experiments/bx3_costreg_timing/time_variants.py > ${g}_bxc_timing.txt 2>&1; cat ${g}_bxc_timing.txt
for v in shipped bxc_xlate bxc_xlate_b2; do
  if [ $v = shipped ]; then unset ADAMVS_LIB_PATH; else export ADAMVS_LIB_PATH=$PWD/ada-mvs_amd/libadamvs_hip.$v.so; fi
  python bench.py --workload cfg2 --batch 128 --precision bf16x3 --no-cascade --no-cpu-baseline --steps 5 > ${g}_bench_cfg2_bx3_$v.json 2> ${g}_bench_cfg2_bx3_$v.err
  echo $v; python tools/show_bench.py ${g}_bench_cfg2_bx3_$v.json | head -4
done
