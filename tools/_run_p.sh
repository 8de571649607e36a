python -m pytest tests/test_hip_parity.py -q -x -k "minimal_filtering or one_role_per_launch or slice_reg_step" 2>&1 | tail -2
python3 bench.py --no-cpu-baseline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 b128', round(d['value'],1), round(d['ms_per_step'],2), d['phase_ms_per_step']['s1.recurrence'])"
for rm in default 1; do
  if [ $rm = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$rm; fi
  python3 bench.py --workload cfg3 --batch 32 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b32 recur mode $rm', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
done
