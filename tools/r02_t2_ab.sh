#!/bin/bash
# tools/r02_t2_ab.sh: transposed CostRegNet2D layers, class-by-class (0) against fused 4-row (1) / 2-row (2) blocks
cd "$(dirname "$0")/.."
for v in 0 1 2; do
  for a in "cfg2 128" "cfg3 4"; do
    set -- $a
    ADAMVS_T2_FUSED=$v python3 bench.py --no-cpu-baseline --workload $1 --batch $2 --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); l=d['cost_reg_layers_ms']
print('T2_FUSED=$v $a: %.2f ms/step; costreg %.2f; conv7 %.3f conv9 %.3f conv11 %.3f' % (d['ms_per_step'], d['phase_ms_per_step']['s1.cost_reg_net_2d'], l['conv7.mode2'], l['conv9.mode2'], l['conv11.mode2']))"
  done
done
