#!/bin/bash
# tools/r02_ab2.sh <tag>: schedules x slot policies, whole-step time only
tag=$1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfgargs in "cfg2 128 fp32" "cfg3 4 fp32" "cfg3 32 fp32" "cfg3 32 bf16x3" "cfg3 4 bf16x3"; do
  set -- $cfgargs
  for mode in 0 1 2; do
    for pol in 0 1; do
      [ $mode = 0 ] && [ $pol = 1 ] && continue
      [ $3 = bf16x3 ] && [ $mode = 2 ] && continue
      out=gpurun_out/${tag}_$1_b$2_$3_m${mode}p${pol}.json
      ADAMVS_RECUR_MODE=$mode ADAMVS_SLOT_POLICY=$pol timeout 600 python3 bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --workload $1 --batch $2 --precision $3 > $out 2> ${out%.json}.err
      python3 -c "
import json,sys
d=json.loads([l for l in open('$out') if l.startswith('{')][0])
print('%-6s B=%-4s %-7s mode %s policy %s: %8.2f ms/step  %7.1f maps/s' % ('$1','$2','$3','$mode','$pol',d['ms_per_step'],d['value']))"
    done
  done
done
