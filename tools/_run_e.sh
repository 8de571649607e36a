for m in default 5 3 1; do
  if [ $m = default ]; then unset ADAMVS_RECUR_MODE; else export ADAMVS_RECUR_MODE=$m; fi
  python3 bench.py --workload cfg3 --batch 4 --no-cpu-baseline --no-cascade --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('cfg4share mode $m', round(d['ms_per_step'],3), {k:v for k,v in p.items() if 'recurrence' in k})"
done
export ADAMVS_RECUR_MODE=5
BATCH=4 tools/small_batch_prof.sh r04b_mode5_fp32 fp32 > /dev/null 2>&1; head -8 gpurun_out/r04b_mode5_fp32_gaps.txt
