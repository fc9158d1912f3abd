#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
T=r04r
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "two_part or row_stats or folded or ln_fold or layernorm_fold" -s 2>&1 | grep -E "two-part|passed|failed|Error|assert" > gpurun_out/${T}_kernel_tests.log
for rep in 1 2; do
  for v in 0 1; do
    echo "VX_STATS_PARTS=$v rep $rep" >> gpurun_out/${T}_ab_stats_parts.txt
    VX_STATS_PARTS=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/${T}_bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab_stats_parts.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py -x -q > gpurun_out/${T}_model_tests.log 2>&1
