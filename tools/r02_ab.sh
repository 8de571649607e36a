#!/bin/bash
# A/B of the recurrence schedules on the GPU box: tools/r02_ab.sh <tag> -> gpurun_out/<tag>_*.json
tag=$1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name, mode, args...
  name=$1; mode=$2; shift 2
  ADAMVS_RECUR_MODE=$mode timeout 600 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > gpurun_out/${tag}_${name}_mode${mode}.json 2> gpurun_out/${tag}_${name}_mode${mode}.err
  python3 tools/show_bench.py gpurun_out/${tag}_${name}_mode${mode}.json | head -3
}
for mode in 2 1 0; do
  run cfg2_b128 $mode --workload cfg2 --batch 128
  run cfg3_b4 $mode --workload cfg3 --batch 4
  run cfg3_b32 $mode --workload cfg3 --batch 32
  run cfg3_b32_bx3 $mode --workload cfg3 --batch 32 --precision bf16x3
  run cfg3_b4_bx3 $mode --workload cfg3 --batch 4 --precision bf16x3
done
