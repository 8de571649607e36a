mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -4
for cfg in "8 1" "16 1" "32 1" "16 2" "16 4" "32 4" "32 8"; do set -- $cfg; python bench.py --steps 3 --warmup 1 --batch $1 --groups $2 --no-cpu-baseline > gpurun_out/b3_$1_$2.log 2>&1; python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/b3_$1_$2.log").read().strip().splitlines()[-1])
    print("B=$1 G=$2", round(r["value"],1), "maps/s", round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phase_ms_per_step"].items()})
except Exception as e:
    print("B=$1 G=$2 FAILED", e); print(open("gpurun_out/b3_$1_$2.log").read()[-1500:])
PY
done
