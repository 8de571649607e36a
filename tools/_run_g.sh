python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -q -x -k "bf16x3" 2>&1 | tail -4
python -m pytest tests/test_msrednet.py -q -x -k "benchmark_shape" 2>&1 | tail -4
python3 bench.py --workload cfg3 --batch 32 --precision bf16x3 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg3 b32 bf16x3', round(d['ms_per_step'],2), {k:v for k,v in p.items() if 'recurrence' in k})"
python3 bench.py --workload cfg3 --batch 4 --precision bf16x3 --no-cpu-baseline --no-roofline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4share bf16x3', d['ms_per_step'])"
python3 bench.py --workload cfg3 --batch 4 --no-cpu-baseline --no-roofline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4share fp32', d['ms_per_step'])"
