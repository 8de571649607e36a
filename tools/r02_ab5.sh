#!/bin/bash
# tools/r02_ab5.sh <tag>: bf16x3 with both GRU levels fused -- four launches per hypothesis (mode 0) against two (mode 1)
tag=$1
cd "$(dirname "$0")/.."
for cfgargs in "cfg3 4" "cfg3 8" "cfg3 16" "cfg3 32" "cfg2 16" "cfg1 32"; do
  set -- $cfgargs
  for mode in 0 1 5; do
    out=gpurun_out/${tag}_$1_b$2_m${mode}.json
    ADAMVS_RECUR_MODE=$mode timeout 600 python3 bench.py --no-cpu-baseline --precision bf16x3 --steps 4 --warmup 2 --workload $1 --batch $2 > $out 2> ${out%.json}.err
    python3 -c "
import json
d=json.loads([l for l in open('$out') if l.startswith('{')][0])
ph=d.get('phase_ms_per_step',{})
print('%-5s B=%-3s bf16x3 mode %s: %8.2f ms/step  %7.1f maps/s   recurrence %s' % ('$1','$2','$mode',d['ms_per_step'],d['value'],' '.join('%s=%.2f'%(k.split('.')[0],v) for k,v in ph.items() if 'recurrence' in k)))" || tail -3 ${out%.json}.err
  done
done
