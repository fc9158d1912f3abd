#!/bin/bash
# GEGLU epilogue: plain vs non-temporal output stores (tools/ringlibs/gnt0.so / gnt1.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2; do
for shape in "L0 geglu" "L1 geglu" "L2 geglu"; do
  for m in 0 1; do
    printf "nt=%s " $m
    timeout 60 tools/gemm_bench tools/ringlibs/gnt$m.so 10 "$shape" 2>&1 | grep "^L[012] " | cut -c1-100
  done
done
done
