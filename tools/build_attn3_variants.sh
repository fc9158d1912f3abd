#!/bin/bash
# one attention-only library per attn3 ablation mask (tools/exp_attn3.sh); tools/attnlibs/ is git-ignored but ships to the box
cd "$(dirname "$0")/../v-express_amd/csrc"
mkdir -p ../../tools/attnlibs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -shared"
for m in 0 1 2 3 4 7 8 16 24 32 63; do
  /opt/rocm/bin/hipcc $F -DVX_ATTN3_ABLATE=$m vx_attn.hip vx_attn3.hip -x hip vx_api.cpp -o ../../tools/attnlibs/abl$m.so &
done
wait
ls -la ../../tools/attnlibs
