#!/bin/bash
# The measurement set of a round, on the GPU box:   tools/round_final.sh r03 [quick]
#   gpurun_out/<r>z_*: GPU test tail, the default bench line (headline + cascade key), kernel stats / stamped traffic / SQ
#   counters of cfg2 (256 tiles: the default batch since round 5) and cfg3 (32 tiles, both precisions), one bench line per other configuration.
# tools/round_collect.sh <r> copies them to their tracked names under profiles/.   "quick": skip the long tail of bench lines.
r=$1; quick=$2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
g=gpurun_out/${r}z
python -m pytest tests -m gpu -q 2>&1 | tail -4 > ${g}_gpu_tests.txt; cat ${g}_gpu_tests.txt
# cfg2 (the headline): kernel stats, stamped traffic, SQ counters, then the default bench line that quotes the traffic
WORKLOAD=cfg2 TILES=256 tools/profile_round.sh ${r}z_cfg2_b256 --workload cfg2 --batch 256 --no-cascade > ${g}_profile_cfg2.log 2>&1
cp ${g}_cfg2_b256_traffic.json profiles/${r}_traffic_cfg2_b256_fp32.json
tools/bench_pmc.sh ${r}z_cfg2_b256 --workload cfg2 --batch 256 --no-cascade > ${g}_pmc_cfg2.log 2>&1
python bench.py > ${g}_bench_default.json 2> ${g}_bench_default.err
python tools/show_bench.py ${g}_bench_default.json
# cfg3 at 32 tiles per step, both precisions
WORKLOAD=cfg3 TILES=32 tools/profile_round.sh ${r}z_cfg3_fp32_b32 --workload cfg3 --batch 32 --no-cascade > ${g}_profile_cfg3_fp32.log 2>&1
WORKLOAD=cfg3 TILES=32 PRECISION=bf16x3 tools/profile_round.sh ${r}z_cfg3_bf16x3_b32 --workload cfg3 --batch 32 --precision bf16x3 --no-cascade > ${g}_profile_cfg3_bf16x3.log 2>&1
tools/bench_pmc.sh ${r}z_cfg3_fp32_b32 --workload cfg3 --batch 32 --no-cascade > ${g}_pmc_cfg3_fp32.log 2>&1
tools/bench_pmc.sh ${r}z_cfg3_bf16x3_b32 --workload cfg3 --batch 32 --precision bf16x3 --no-cascade > ${g}_pmc_cfg3_bf16x3.log 2>&1
[ -n "$quick" ] && exit 0
run() { name=$1; shift; timeout 900 python bench.py --no-cascade "$@" > ${g}_bench_$name.json 2> ${g}_bench_$name.err; python tools/show_bench.py ${g}_bench_$name.json | head -2; }
run cfg2_bf16x3 --precision bf16x3
run cfg3_fp32_b128 --workload cfg3 --batch 128 --no-cpu-baseline
run cfg3_bf16x3_b128 --workload cfg3 --batch 128 --precision bf16x3 --no-cpu-baseline
run cfg5_fp32_b8 --workload cfg5 --batch 8 --no-cpu-baseline
run cfg5_bf16x3_b8 --workload cfg5 --batch 8 --precision bf16x3 --no-cpu-baseline
run cfg1_fp32_b128 --workload cfg1 --batch 128
run msrednet_cfg3_b1 --model msrednet --workload cfg3 --no-cpu-baseline
run msrednet_cfg3_b16 --model msrednet --workload cfg3 --red-batch 16 --no-cpu-baseline
