#!/bin/bash
# libraries with compile-time variants of the fused GroupNorm + SiLU + 3x3 convolution (tools/conv3_bench.py via VX_LIBRARY):
#   build_conv3_variants.sh "name:flag,flag name2:flag ..."      e.g. "abl2:-DVX_C3_ABLATE=2 abl4:-DVX_C3_ABLATE=4"
# (VX_C3_ABLATE: 1 no MFMA, 2 no plane normalisation, 4 no plane copies after the prologue, 8 no weight copies)
cd "$(dirname "$0")/../v-express_amd/csrc"
mkdir -p ../../tools/c3libs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-unused-function -c"
for v in $1; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc $F ${flags//,/ } vx_conv3.hip -o /tmp/c3_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC vx_gemm.o vx_gemm_ring.o /tmp/c3_$name.o vx_ff.o vx_tblock.o vx_norm.o vx_attn.o vx_attn3.o vx_elem.o vx_api.o -o ../../tools/c3libs/$name.so ) &
done
wait
ls -la ../../tools/c3libs
