for m in 7 15; do
  export ADAMVS_GRU_WINO=$m
  python -m pytest tests/test_hip_parity.py -q -x -k "slice_reg_step_golden or slice_reg_step_ragged or slice_reg_step_many" 2>&1 | tail -1
  ADAMVS_RECUR_MODE=0 python3 bench.py --batch 128 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg2 b128 wino mask $m', round(d['ms_per_step'],2), p['s1.recurrence'])"
done
export ADAMVS_GRU_WINO=7
for rm in default 0; do
  if [ $rm = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$rm; fi
  python3 bench.py --workload cfg3 --batch 32 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b32 recur mode $rm', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
  python3 bench.py --workload cfg3 --batch 8 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b8 recur mode $rm', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
done
unset ADAMVS_RECUR_MODE
python3 bench.py --workload cfg5 --batch 8 --no-cpu-baseline --no-cascade --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg5 b8', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
