#!/bin/bash
# Geometry sensitivity of the plane sweeps, on the GPU box: bench.py at rig baselines 8 (SURVEY 8c's recipe), 32, 128, 512.
#   tools/geometry_sweep.sh <tag> [bench.py args, default: the cfg2 headline]  -> gpurun_out/<tag>_baseline<X>.json + a table
# Per baseline: maps/s, the pair_similarity and aggregate_conv1 phase times, parity of tile 0 against the oracle.
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tab=gpurun_out/${tag}_geometry.txt
: > $tab
for X in ${BASELINES:-8 32 128 512}; do
  out=gpurun_out/${tag}_baseline$X
  timeout 900 python3 bench.py --no-cascade --baseline $X "$@" > $out.json 2> $out.err || tail -3 $out.err
  python3 - $out.json $X >> $tab <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d.get("phase_ms_per_step", {})
par = d.get("parity_rel_l1", {})
print("baseline %5s  %8.1f maps/s  %8.2f ms/step  %s  parity depth %.2e conf %.2e (%.1e intervals)" % (
    sys.argv[2], d["value"], d["ms_per_step"], "  ".join("%s %.2f" % (k, v) for k, v in ph.items() if "pair_sim" in k or "aggregate" in k),
    par.get("depth", float("nan")), par.get("photometric_confidence", float("nan")), par.get("depth_abs_err_in_finest_intervals", float("nan"))))
PY
done
cat $tab
