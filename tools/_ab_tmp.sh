mkdir -p gpurun_out
timeout 300 python tools/wino_bench.py --time-only > gpurun_out/wino_fill1.txt 2>&1
ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.fill0.so timeout 300 python tools/wino_bench.py --time-only > gpurun_out/wino_fill0.txt 2>&1
timeout 300 python tools/wino_bench.py --time-only > gpurun_out/wino_fill1b.txt 2>&1
grep "F(2x2" gpurun_out/wino_fill1.txt gpurun_out/wino_fill0.txt gpurun_out/wino_fill1b.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wino or winograd or cost_reg" 2>&1 | tail -2
