#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for v in "" pin; do
  echo "=== variant '$v'" >> gpurun_out/r04m_debug.txt
  if [ -n "$v" ]; then export VX_LIBRARY=$PWD/tools/tblibs/$v.so; fi
  timeout 300 python tools/tb_debug.py 1 8 2>&1 | grep -E "probe|O \(wo" >> gpurun_out/r04m_debug.txt
done
