for m in default 6; do
  if [ $m = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$m; fi
  python3 bench.py --workload cfg3 --batch 32 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b32 mode $m', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
done
