#!/bin/bash
# A/B of bench.py under environment variants, on the GPU box (replaces the one-off r02_*.sh scripts of round 2; those are
# in the history at c4fe1dc).
#
#   tools/ab.sh <tag> "<env A>|<env B>|..." "<case>=<bench.py args>" ["<case>=<args>" ...]
#
# e.g. the recurrence schedules of round 2:
#   tools/ab.sh sched "ADAMVS_RECUR_MODE=0|ADAMVS_RECUR_MODE=1|ADAMVS_RECUR_MODE=3" \
#        "cfg2_b128=--workload cfg2 --batch 128" "cfg3_b4=--workload cfg3 --batch 4" "cfg3_b32_bx3=--workload cfg3 --batch 32 --precision bf16x3"
# An empty variant ("|ADAMVS_X=1") is the unmodified build.  A variant may also name another library build:
# ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.<name>.so (tools/build_variant.py).  Lines land in gpurun_out/<tag>_<case>_<variant>.json;
# the summary (value, ms per step, phase table) is printed per run.  EXTRA="--steps 10" adds bench arguments to every run.
tag=$1; variants=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
IFS='|' read -ra VARS <<< "$variants"
[ ${#VARS[@]} -eq 0 ] && VARS=("")
for v in "${VARS[@]}"; do
  vname=$(echo "$v" | sed 's/ADAMVS_//g; s#[ /]#_#g; s/=/-/g'); [ -z "$v" ] && vname=base
  for case in "$@"; do
    name=${case%%=*}; args=${case#*=}
    out=gpurun_out/${tag}_${name}_${vname}
    echo "== $name [$v]"
    env $v timeout 900 python3 bench.py --no-cpu-baseline --no-cascade --steps 5 --warmup 2 $EXTRA $args > $out.json 2> $out.err || tail -3 $out.err
    python3 tools/show_bench.py $out.json | head -${SHOW:-3}
  done
done
