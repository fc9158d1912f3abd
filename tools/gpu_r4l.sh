#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for v in "" keepwf pin both; do
  echo "=== variant '$v'" >> gpurun_out/r04l_debug.txt
  if [ -n "$v" ]; then export VX_LIBRARY=$PWD/tools/tblibs/$v.so; fi
  timeout 300 python tools/tb_debug.py 1 8 2>&1 | grep -E "rel-L2|by pixel|by head" >> gpurun_out/r04l_debug.txt
  timeout 300 python tools/tb_bench.py 40 2>&1 | grep fused | tail -1 >> gpurun_out/r04l_debug.txt
done
