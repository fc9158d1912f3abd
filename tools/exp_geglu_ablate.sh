#!/bin/bash
# ring GEGLU epilogue ablation: 0 = product, 64 = no GELU (value * gate), 32 = no epilogue stores, 96 = both,
# 24 = no A/B copies (MFMA + epilogue only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for shape in "L0 geglu" "L1 geglu" "L2 geglu"; do
  for m in 0 64 32 96 24; do
    printf "mask %-3s " $m
    timeout 60 tools/gemm_bench tools/ringlibs/gabl$m.so 10 "$shape" 2>&1 | grep "^L[012] " | cut -c1-80
  done
done
