cd /root/repo
for g2 in 6 9 12 16 22; do
  for b in 4 8; do
    ADAMVS_RECUR_COSTS="2.85,4.07,2.39,9.98,5.53,4.46,13.8,3.2,$g2,5.0,0.6,0.8,0.42" python3 bench.py --no-cpu-baseline --precision bf16x3 --steps 4 --warmup 2 --workload cfg3 --batch $b 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ph=d['phase_ms_per_step']
print('g2bx=$g2 B=$b: %.2f ms/step; ' % d['ms_per_step'] + ' '.join('%s=%.2f' % (k.split('.')[0], x) for k, x in ph.items() if 'recurrence' in k))"
  done
done
