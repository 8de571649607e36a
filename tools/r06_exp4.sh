#!/bin/bash
cd "$(dirname "$0")/.."
g=gpurun_out/r06f
python -m pytest tests -m gpu -q 2>&1 | tail -15 > ${g}_gpu_tests.txt; cat ${g}_gpu_tests.txt
python bench.py --workload cfg3 --batch 32 --precision bf16x3 --no-cascade --no-cpu-baseline --steps 5 > ${g}_bench_cfg3_bx3_b32.json 2> ${g}_bench_cfg3_bx3_b32.err; python tools/show_bench.py ${g}_bench_cfg3_bx3_b32.json | head -3
python tools/predict_size_prof.py > ${g}_predict_size.txt 2>&1; tail -15 ${g}_predict_size.txt
