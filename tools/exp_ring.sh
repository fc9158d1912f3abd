#!/bin/bash
# ring kernel vs classic kernels (both with inline-asm LDS DMA)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=v-express_amd/libvexpress_hip.so
{
echo "=== classic kernels (VX_GEMM_RING=0)"
VX_GEMM_RING=0 timeout 120 tools/gemm_bench $L 20
echo "=== ring kernel where eligible"
for f in "L0 lin" "L0 ffout" "prepad" "L1 lin" "L1 ffout"; do
  timeout 60 tools/gemm_bench $L 20 "$f" | grep -v "^norm\|^L. \(320\|640\|1280\|cat\)\|^VAE 512^2 128 x4\|^shape\|^weighted"
  echo "rc=$?"
done
} > gpurun_out/ring.txt 2>&1
tail -30 gpurun_out/ring.txt
