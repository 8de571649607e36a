#!/bin/bash
# tools/r02_bench_lines.sh: the bench lines of tools/r02_final.sh without the rocprof passes (after a change that leaves the kernels alone)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r02z_gpu_tests.txt; cat gpurun_out/r02z_gpu_tests.txt
run() { name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r02z_bench_$name.json 2> gpurun_out/r02z_bench_$name.err; python tools/show_bench.py gpurun_out/r02z_bench_$name.json | head -2; }
run default
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
python bench.py --workload cfg3 --batch 32 > gpurun_out/r02z_cfg3_fp32_b32_bench.json 2>/dev/null
python bench.py --workload cfg3 --batch 32 --precision bf16x3 > gpurun_out/r02z_cfg3_bf16x3_b32_bench.json 2>/dev/null
