#!/bin/bash
# round 4, GPU call D: fused feed-forward, eight waves (two per SIMD) vs four waves (one per SIMD), + ablations of the eight-wave form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
T=r04d
{
for w in 8 4 8 4; do
  echo "=== product library, VX_FF_WAVES=$w"
  VX_FF_WAVES=$w timeout 100 tools/ff_bench v-express_amd/libvexpress_hip.so 20
done
for lib in tools/fflibs/abl1.so tools/fflibs/abl2.so tools/fflibs/abl6.so tools/fflibs/abl24.so tools/fflibs/abl63.so; do
  echo "=== $lib (eight waves)"
  timeout 100 tools/ff_bench $lib 20 | tail -2
done
} > $OUT/${T}_ff_variants.txt 2>&1
cat $OUT/${T}_ff_variants.txt
