#!/bin/bash
# one GEMM + FF library per ablation mask of the fused feed-forward prototype (tools/ff_bench)
cd "$(dirname "$0")/../v-express_amd/csrc"
mkdir -p ../../tools/fflibs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-unused-function -shared"
for m in ${1:-1 2 4 6 8 16 24 32 63}; do
  /opt/rocm/bin/hipcc $F -DVX_FF_ABLATE=$m vx_gemm.hip vx_gemm_ring.hip vx_ff.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/fflibs/abl$m.so &
done
# schedule experiments (second argument: list of "name:flags")
for v in ${2:-}; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F ${flags//,/ } vx_gemm.hip vx_gemm_ring.hip vx_ff.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/fflibs/$name.so &
done
wait
ls -la ../../tools/fflibs
