#!/bin/bash
# k-major form of the 8-phase kernel: parity tests, then dW timing A/B against the 128 x 128 kernel on the same box
mkdir -p gpurun_out/r4m
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "kmajor or matmul_tn or matmul_nn" > gpurun_out/r4m/pytest.log 2>&1 || { tail -30 gpurun_out/r4m/pytest.log; exit 1; }
tail -3 gpurun_out/r4m/pytest.log
echo "== KM8=1" && timeout -k 10 300 python tools/tn_microbench.py 2>&1 | grep dW | tee gpurun_out/r4m/tn_km8_1.txt &&
echo "== KM8=0" && AVAMD_GEMM_KM8=0 timeout -k 10 300 python tools/tn_microbench.py 2>&1 | grep dW | tee gpurun_out/r4m/tn_km8_0.txt
