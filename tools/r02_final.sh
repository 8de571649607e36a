#!/bin/bash
# Final measurements of the round on the GPU box: tools/r02_final.sh  -> gpurun_out/r02z_*
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r02z_gpu_tests.txt; cat gpurun_out/r02z_gpu_tests.txt
# cfg2 (the headline): kernel stats, stamped traffic, SQ counters, then the bench line that quotes the traffic
WORKLOAD=cfg2 TILES=128 tools/profile_round.sh r02z_cfg2_b128 --workload cfg2 --batch 128 > gpurun_out/r02z_profile_cfg2.log 2>&1
cp gpurun_out/r02z_cfg2_b128_traffic.json profiles/r02_traffic_cfg2_b128_fp32.json
tools/bench_pmc.sh r02z_cfg2_b128 --workload cfg2 --batch 128 > gpurun_out/r02z_pmc_cfg2.log 2>&1
python bench.py > gpurun_out/r02z_bench_default.json 2> gpurun_out/r02z_bench_default.err
python tools/show_bench.py gpurun_out/r02z_bench_default.json
# cfg3 at 32 tiles per step, both precisions
WORKLOAD=cfg3 TILES=32 tools/profile_round.sh r02z_cfg3_fp32_b32 --workload cfg3 --batch 32 > gpurun_out/r02z_profile_cfg3_fp32.log 2>&1
WORKLOAD=cfg3 TILES=32 PRECISION=bf16x3 tools/profile_round.sh r02z_cfg3_bf16x3_b32 --workload cfg3 --batch 32 --precision bf16x3 > gpurun_out/r02z_profile_cfg3_bf16x3.log 2>&1
tools/bench_pmc.sh r02z_cfg3_fp32_b32 --workload cfg3 --batch 32 > /dev/null 2>&1
tools/bench_pmc.sh r02z_cfg3_bf16x3_b32 --workload cfg3 --batch 32 --precision bf16x3 > /dev/null 2>&1
run() { name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r02z_bench_$name.json 2> gpurun_out/r02z_bench_$name.err; python tools/show_bench.py gpurun_out/r02z_bench_$name.json | head -2; }
run cfg2_bf16x3 --precision bf16x3
run cfg3_fp32_b128 --workload cfg3 --batch 128 --no-cpu-baseline
run cfg3_bf16x3_b128 --workload cfg3 --batch 128 --precision bf16x3 --no-cpu-baseline
run cfg4_share_fp32_b4 --workload cfg3 --batch 4
run cfg4_share_bf16x3_b4 --workload cfg3 --batch 4 --precision bf16x3
run cfg5_fp32_b4 --workload cfg5 --batch 4 --no-cpu-baseline
run cfg5_bf16x3_b4 --workload cfg5 --batch 4 --precision bf16x3 --no-cpu-baseline
run cfg5_fp32_b8 --workload cfg5 --batch 8 --no-cpu-baseline
run cfg5_bf16x3_b8 --workload cfg5 --batch 8 --precision bf16x3 --no-cpu-baseline
run cfg1_fp32_b128 --workload cfg1 --batch 128
run msrednet_cfg3_b1 --model msrednet --workload cfg3 --no-cpu-baseline
run msrednet_cfg3_b16 --model msrednet --workload cfg3 --red-batch 16 --no-cpu-baseline
