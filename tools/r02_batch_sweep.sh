#!/bin/bash
# tools/r02_batch_sweep.sh <tag>: cfg3 per-tile time against tiles per step
tag=$1
cd "$(dirname "$0")/.."
for prec in fp32 bf16x3; do
  for b in 16 32 64 96 128; do
    out=gpurun_out/${tag}_cfg3_b${b}_${prec}.json
    timeout 900 python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 --workload cfg3 --batch $b --precision $prec > $out 2> ${out%.json}.err
    python3 -c "
import json
d=json.loads([l for l in open('$out') if l.startswith('{')][0])
print('cfg3 B=%-4s %-7s %8.2f ms/step  %7.1f maps/s  %.3f ms/tile' % ('$b','$prec',d['ms_per_step'],d['value'],d['ms_per_step']/$b))" || tail -3 ${out%.json}.err
  done
done
