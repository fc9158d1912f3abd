#!/bin/bash
# round 4 J: the fused temporal attention block - kernel tests, microbenchmark, whole-path A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=r04j
rocm-smi --showclocks 2>/dev/null | head -12 > gpurun_out/${T}_box.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tblock" -s > gpurun_out/${T}_kernel_tests.log 2>&1
echo "tblock tests exit $?" >> gpurun_out/${T}_kernel_tests.log
timeout 300 python tools/tb_bench.py 40 > gpurun_out/${T}_tb_bench.txt 2>&1
for rep in 1 2; do
  for v in 0 1; do
    echo "VX_TB_FUSED=$v rep $rep" >> gpurun_out/${T}_ab_tblock.txt
    VX_TB_FUSED=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/${T}_bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab_tblock.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s > gpurun_out/${T}_fullsize_tests.log 2>&1
echo "fullsize exit $?" >> gpurun_out/${T}_fullsize_tests.log
