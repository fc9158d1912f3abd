#!/bin/bash
# round 4, GPU call G: fused feed-forward v3 (stage 2 as 2 x 4, next tile's x prefetched) vs the r04e form, deeper W1 prefetch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
T=r04g
{
for lib in v-express_amd/libvexpress_hip.so tools/fflibs/base.so tools/fflibs/s2only.so tools/fflibs/xpfonly.so tools/fflibs/pf5.so tools/fflibs/pf9.so v-express_amd/libvexpress_hip.so tools/fflibs/base.so; do
  echo "=== $lib"
  timeout 100 tools/ff_bench $lib 20
done
} > $OUT/${T}_ff_variants.txt 2>&1
grep -v "round 0" $OUT/${T}_ff_variants.txt
