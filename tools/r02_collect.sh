#!/bin/bash
# tools/r02_collect.sh: copy what tools/r02_final.sh left under gpurun_out/r02z_* to the tracked names under profiles/
cd "$(dirname "$0")/.."
g=gpurun_out/r02z
cp ${g}_bench_default.json profiles/r02_bench_cfg2_fp32.json
cp ${g}_cfg2_b128_kernel_stats.csv profiles/r02_kernel_stats_cfg2_b128_fp32.csv
cp ${g}_cfg2_b128_traffic.json profiles/r02_traffic_cfg2_b128_fp32.json
cp ${g}_cfg2_b128_traffic.txt profiles/r02_hbm_traffic_pmc_cfg2_b128_fp32.txt
cp ${g}_cfg2_b128_sq_counters.txt profiles/r02_sq_counters_cfg2_b128_fp32.txt
grep "MFMA pipe busy" ${g}_pmc_cfg2.log > profiles/r02_mfma_busy_cfg2_b128_fp32.txt
for p in fp32 bf16x3; do
  cp ${g}_cfg3_${p}_b32_kernel_stats.csv profiles/r02_kernel_stats_cfg3_b32_${p}.csv
  cp ${g}_cfg3_${p}_b32_traffic.json profiles/r02_traffic_cfg3_b32_${p}.json
  cp ${g}_cfg3_${p}_b32_traffic.txt profiles/r02_hbm_traffic_pmc_cfg3_b32_${p}.txt
  cp ${g}_cfg3_${p}_b32_sq_counters.txt profiles/r02_sq_counters_cfg3_b32_${p}.txt
  cp ${g}_cfg3_${p}_b32_bench.json profiles/r02_bench_cfg3_${p}_b32.json
done
for n in cfg2_bf16x3 cfg3_fp32_b128 cfg3_bf16x3_b128 cfg4_share_fp32_b4 cfg4_share_bf16x3_b4 cfg5_fp32_b4 cfg5_bf16x3_b4 cfg5_fp32_b8 cfg5_bf16x3_b8 cfg1_fp32_b128 msrednet_cfg3_b1 msrednet_cfg3_b16; do
  cp ${g}_bench_$n.json profiles/r02_bench_$n.json
done
cp ${g}_gpu_tests.txt profiles/r02_gpu_tests.txt
cp gpurun_out/parity_full_size.jsonl profiles/r02_parity_full_size.jsonl
