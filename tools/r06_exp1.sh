#!/bin/bash
# round 6, GPU batch 1: the whole GPU suite on the new tree, then the A/B of the 96-channel F(2x2, 3x3) groups (option wino_mt6)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
g=gpurun_out/r06b
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > ${g}_gpu_tests.txt; cat ${g}_gpu_tests.txt
for v in 0 1; do
  ADAMVS_WINO_MT6=$v python tools/wino_bench.py --time-only > ${g}_wino_bench_mt6_$v.txt 2>&1; tail -12 ${g}_wino_bench_mt6_$v.txt
  ADAMVS_WINO_MT6=$v python bench.py --no-cascade --no-cpu-baseline --steps 5 > ${g}_bench_cfg2_mt6_$v.json 2> ${g}_bench_cfg2_mt6_$v.err
  python tools/show_bench.py ${g}_bench_cfg2_mt6_$v.json | head -30
done
