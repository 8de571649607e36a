python -m pytest tests/test_hip_parity.py -q -x -k "pipelined or minimal_filtering or one_role_per_launch or slice_reg_step" 2>&1 | tail -3
python -m pytest tests/test_full_size_parity.py -q -x 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 b128', round(d['value'],1), round(d['ms_per_step'],2), d['phase_ms_per_step'], {k:v for k,v in d['roofline'].items() if 'frac' in k})"
