#!/bin/bash
# A/B of the stride-2 CostRegNet2D layers on the GPU box: the direct kernel (ADAMVS_S2_PAIRS=0) against the pair form (round 5).
cd "$(dirname "$0")/.."
for v in 0 1; do
  echo "== ADAMVS_S2_PAIRS=$v"
  ADAMVS_S2_PAIRS=$v python3 bench.py --no-cpu-baseline --no-cascade --steps 5 --warmup 2 > gpurun_out/s2ab_$v.json 2> gpurun_out/s2ab_$v.err
  python3 tools/show_bench.py gpurun_out/s2ab_$v.json | head -5
done
