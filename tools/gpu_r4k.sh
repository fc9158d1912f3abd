#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 300 python tools/tb_debug.py 1 8 > gpurun_out/r04k_debug.txt 2>&1
timeout 300 python tools/tb_debug.py 2 64 >> gpurun_out/r04k_debug.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tblock" -s 2>&1 | grep -E "tblock_fused b=|passed|failed" > gpurun_out/r04k_kernel_tests.log
timeout 300 python tools/tb_bench.py 40 > gpurun_out/r04k_tb_bench.txt 2>&1
