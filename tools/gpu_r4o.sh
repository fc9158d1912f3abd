#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
T=r04o
echo "product" > gpurun_out/${T}_tb_variants.txt
timeout 120 python tools/tb_bench.py 40 2>&1 | grep -E "fused|three" | tail -2 >> gpurun_out/${T}_tb_variants.txt
for v in $(ls tools/tblibs 2>/dev/null | sed 's/\.so//'); do
  echo "== $v" >> gpurun_out/${T}_tb_variants.txt
  VX_LIBRARY=$PWD/tools/tblibs/$v.so timeout 120 python tools/tb_bench.py 40 2>&1 | grep fused | tail -1 >> gpurun_out/${T}_tb_variants.txt
done
