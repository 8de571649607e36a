python -m pytest tests/test_hip_parity.py -q -x -k "pipelined or minimal_filtering or one_role_per_launch" 2>&1 | tail -3
python -m pytest tests/test_full_size_parity.py -q -x -k "cfg4" 2>&1 | tail -3
for m in 0 7; do
export ADAMVS_GRU_WINO=$m
python3 bench.py --workload cfg3 --batch 4 --no-cpu-baseline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg4share wino $m', round(d['ms_per_step'],3), {k:v for k,v in p.items() if 'recurrence' in k})"
python3 bench.py --workload cfg3 --batch 8 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b8 wino $m', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
python3 bench.py --workload cfg3 --batch 16 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b16 wino $m', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
done
