#!/bin/bash
# round 4, GPU call C: fused feed-forward v2 (three balanced MFMA segments) vs v1, and the ablation variants of v2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
T=r04c
{
for lib in v-express_amd/libvexpress_hip.so tools/fflibs/v1.so tools/fflibs/abl1.so tools/fflibs/abl2.so tools/fflibs/abl6.so tools/fflibs/abl24.so tools/fflibs/abl63.so v-express_amd/libvexpress_hip.so; do
  echo "=== $lib"
  timeout 100 tools/ff_bench $lib 20
done
} > $OUT/${T}_ff_variants.txt 2>&1
cat $OUT/${T}_ff_variants.txt
