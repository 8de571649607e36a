#!/bin/bash
cd "$(dirname "$0")/.."
g=gpurun_out/r06c
python tools/dbg_graph.py > ${g}_dbg_graph.txt 2>&1; cat ${g}_dbg_graph.txt | tail -30
python -m pytest tests -m gpu -q --deselect tests/test_hip_parity.py::test_graphed_forward_equals_eager_for_every_depth_range 2>&1 | tail -15 > ${g}_gpu_tests.txt; cat ${g}_gpu_tests.txt
