mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf16x3 or precision or schedule or pipelin or recurrence" > gpurun_out/c2r_tests.log 2>&1; tail -3 gpurun_out/c2r_tests.log
timeout 200 tools/step_prof.sh c2r --workload cfg2 --batch 128 --precision bf16x3 --iters 30 2>&1 | grep -E "gru1|gru2|us per step"
timeout 300 python bench.py --workload cfg3 --batch 32 --precision bf16x3 --no-cascade --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload cfg3 --batch 128 --precision bf16x3 --no-cascade --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b128', d['value'], d['ms_per_step'])"
