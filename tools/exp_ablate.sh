#!/bin/bash
# GEMM ablation experiment (tools/gemm_bench + the -DVX_ABLATE library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=v-express_amd/libvexpress_hip_ablate.so
export ABLATE=0,1,2,3,4,8,16,20,23
{
for f in "L0 lin" "L0 geglu" "L0 qkv" "L0 ffout" "L1 lin" "L1 geglu" "L2 lin" "L0 conv3x3 320>320 prepad" "L1 conv3x3 640>640 prepad" "L2 conv3x3 1280>1280 prepad"; do
  tools/gemm_bench $L 20 "$f" | grep -v "^norm\|^L. \(320\|640\|1280\|cat\)\|^VAE 512^2 128 x4\|^shape\|^weighted"
done
echo "=== small tile"
for f in "L0 lin" "L0 geglu" "L0 qkv" "L0 ffout" "L1 lin"; do
  VX_GEMM_TILE=small ABLATE=0,1,3 tools/gemm_bench $L 20 "$f" | grep -v "^norm\|^L. \(320\|640\|1280\|cat\)\|^VAE 512^2 128 x4\|^shape\|^weighted"
done
} > gpurun_out/ablate.txt 2>&1
tail -5 gpurun_out/ablate.txt
