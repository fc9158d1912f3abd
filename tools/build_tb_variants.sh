#!/bin/bash
# libraries with variants of the fused temporal attention block (tools/tb_debug.py / tb_bench.py via VX_LIBRARY):
#   build_tb_variants.sh "name:flag,flag name2:flag ..."
cd "$(dirname "$0")/../v-express_amd/csrc"
mkdir -p ../../tools/tblibs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-unused-function -c"
for v in $1; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc $F ${flags//,/ } vx_tblock.hip -o /tmp/tb_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC vx_gemm.o vx_gemm_ring.o vx_ff.o /tmp/tb_$name.o vx_norm.o vx_attn.o vx_attn3.o vx_elem.o vx_api.o -o ../../tools/tblibs/$name.so ) &
done
wait
ls -la ../../tools/tblibs
