python -m pytest tests/test_hip_parity.py -q -x -k "pipelined or random_shapes" 2>&1 | tail -8
for m in default 5; do
  if [ $m = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$m; fi
  python3 bench.py --workload cfg3 --batch 4 --no-cpu-baseline --no-roofline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4share mode $m', d['ms_per_step'])"
done
for m in default 6; do
  if [ $m = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$m; fi
  python3 bench.py --workload cfg3 --batch 32 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 b32 mode $m', d['ms_per_step'], d['phase_ms_per_step'])"
  python3 bench.py --batch 128 --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 b128 mode $m', d['ms_per_step'], d['phase_ms_per_step'])"
done
