#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
T=r04p
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tblock" -s 2>&1 | grep -E "tblock_fused b=|passed|failed|Error" > gpurun_out/${T}_kernel_tests.log
timeout 300 python tools/tb_bench.py 40 > gpurun_out/${T}_tb_bench.txt 2>&1
for v in $(ls tools/tblibs 2>/dev/null | sed 's/\.so//'); do
  echo "== $v" >> gpurun_out/${T}_tb_bench.txt
  VX_LIBRARY=$PWD/tools/tblibs/$v.so timeout 120 python tools/tb_bench.py 40 2>&1 | grep fused | tail -1 >> gpurun_out/${T}_tb_bench.txt
done
