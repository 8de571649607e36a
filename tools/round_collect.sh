#!/bin/bash
# tools/round_collect.sh r03: copy what tools/round_final.sh left under gpurun_out/<r>z_* to the tracked names under profiles/
r=$1
cd "$(dirname "$0")/.."
g=gpurun_out/${r}z
cp ${g}_bench_default.json profiles/${r}_bench_default.json
cp ${g}_cfg2_b256_kernel_stats.csv profiles/${r}_kernel_stats_cfg2_b256_fp32.csv
cp ${g}_cfg2_b256_traffic.json profiles/${r}_traffic_cfg2_b256_fp32.json
cp ${g}_cfg2_b256_traffic.txt profiles/${r}_hbm_traffic_pmc_cfg2_b256_fp32.txt
cp ${g}_cfg2_b256_sq_counters.txt profiles/${r}_sq_counters_cfg2_b256_fp32.txt
grep "MFMA pipe busy" ${g}_pmc_cfg2.log > profiles/${r}_mfma_busy_cfg2_b256_fp32.txt
for p in fp32 bf16x3; do
  cp ${g}_cfg3_${p}_b32_kernel_stats.csv profiles/${r}_kernel_stats_cfg3_b32_${p}.csv
  cp ${g}_cfg3_${p}_b32_traffic.json profiles/${r}_traffic_cfg3_b32_${p}.json
  cp ${g}_cfg3_${p}_b32_traffic.txt profiles/${r}_hbm_traffic_pmc_cfg3_b32_${p}.txt
  cp ${g}_cfg3_${p}_b32_sq_counters.txt profiles/${r}_sq_counters_cfg3_b32_${p}.txt
  cp ${g}_cfg3_${p}_b32_bench.json profiles/${r}_bench_cfg3_${p}_b32.json
  grep "MFMA pipe busy" ${g}_pmc_cfg3_${p}.log > profiles/${r}_mfma_busy_cfg3_b32_${p}.txt
done
for f in ${g}_bench_*.json; do
  n=$(basename $f .json); n=${n#${r}z_bench_}
  [ "$n" = default ] || cp $f profiles/${r}_bench_$n.json
done
cp ${g}_gpu_tests.txt profiles/${r}_gpu_tests.txt
[ -f gpurun_out/parity_full_size.jsonl ] && cp gpurun_out/parity_full_size.jsonl profiles/${r}_parity_full_size.jsonl
ls profiles/${r}_* | wc -l
