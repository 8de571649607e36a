python -m pytest tests/test_hip_parity.py -q -x -k "slice_reg_step or drop_in_forward" 2>&1 | tail -3
for m in 0 1 2 4 7; do
  export ADAMVS_GRU_WINO=$m
  python -m pytest tests/test_hip_parity.py -q -x -k "slice_reg_step_golden or slice_reg_step_ragged or slice_reg_step_many" 2>&1 | tail -2
  ADAMVS_RECUR_MODE=0 python3 bench.py --batch 128 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg2 b128 wino mask $m', round(d['ms_per_step'],2), p['s1.recurrence'], d.get('parity_rel_l1'))"
done
