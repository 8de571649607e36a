cd /root/repo
for b in 1 2 3 4 8 16 128; do
  python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --workload cfg2 --batch $b 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ph=d['phase_ms_per_step']
print('B=%-3s %.2f ms/step; per tile: aggregate %.3f pair %.3f costreg %.3f recurrence %.3f' % ('$b', d['ms_per_step'], ph['s1.aggregate_conv1']/$b, ph['s1.pair_similarity']/$b, ph['s1.cost_reg_net_2d']/$b, ph['s1.recurrence']/$b))"
done
