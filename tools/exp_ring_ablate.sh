#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for shape in "L0 conv3x3 320>320 prepad" "L1 ffout" "L0 lin"; do
  for m in 0 1 2 4 6 10 18 26 32 34; do
    printf "mask %-3s " $m
    timeout 60 tools/gemm_bench tools/ringlibs/abl$m.so 10 "$shape" 2>&1 | grep "^L[01] " | cut -c1-75
  done
done
