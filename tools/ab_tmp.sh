mkdir -p gpurun_out
b() { timeout 300 python bench.py --workload cfg3 --batch $1 --precision bf16x3 --no-cascade --no-cpu-baseline --steps $2 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 b$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
for rep in 1 2; do
  unset ADAMVS_LIB_PATH; b 32 5 ahead; b 128 3 ahead
  export ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.noah.so; b 32 5 noah; b 128 3 noah
done
