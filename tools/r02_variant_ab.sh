#!/bin/bash
# tools/r02_variant_ab.sh "<variants>" "<bench args>": the recurrence / aggregation phases under experimental builds (tools/build_variant.py)
cd "$(dirname "$0")/.."
for v in $1; do
  if [ "$v" != shipped ]; then export ADAMVS_LIB_PATH=$PWD/ada-mvs_amd/libadamvs_hip.$v.so; else unset ADAMVS_LIB_PATH; fi
  python3 bench.py --no-cpu-baseline --steps 3 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ph=d['phase_ms_per_step']
print('%-9s %s: %.2f ms/step; ' % ('$v', '$2', d['ms_per_step']) + ' '.join('%s=%.2f' % (k, x) for k, x in ph.items() if 'recurrence' in k or 'aggregate' in k))"
done
