#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
T=r04q
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tblock" -s 2>&1 | grep -E "tblock_fused b=|passed|failed|Error|assert" > gpurun_out/${T}_kernel_tests.log
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
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q > gpurun_out/${T}_fullsize_tests.log 2>&1
