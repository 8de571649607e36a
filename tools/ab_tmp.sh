mkdir -p gpurun_out
b() { timeout 300 python bench.py --workload cfg3 --batch $1 --precision bf16x3 --no-cascade --no-cpu-baseline --steps $2 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 b$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for k in 13.8 12.5 11.3 10.3 13.8; do
  export ADAMVS_RECUR_COSTS="2.85,4.07,2.39,9.98,5.53,5.0,$k,3.2,12.0,5.0,0.6,0.8,0.42"
  b 32 5 k1bx=$k; b 4 12 k1bx=$k; b 1 20 k1bx=$k
done
