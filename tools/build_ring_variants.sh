#!/bin/bash
# one GEMM-only library per ring-kernel ablation mask (tools/exp_ring_ablate.sh)
cd "$(dirname "$0")/../v-express_amd/csrc"
mkdir -p ../../tools/ringlibs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -shared"
for m in 0 1 2 4 6 10 18 26 32 34; do
  /opt/rocm/bin/hipcc $F -DVX_RING_ABLATE=$m vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/abl$m.so &
done
for m in 1 64 65; do
  /opt/rocm/bin/hipcc $F -DVX_GEMM_ABLATE=$m vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/cabl$m.so &
done
for m in 0 1 2 3; do   # experimental issue placements (tools/exp_ring_missue.sh)
  /opt/rocm/bin/hipcc $F -DVX_RING_MISSUE=$m vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/mi$m.so &
  /opt/rocm/bin/hipcc $F -DVX_RING_MISSUE=$m -DVX_RING_TRACE vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/mi${m}_trace.so &
done
for p in 0 2; do     # s_setprio placement
  /opt/rocm/bin/hipcc $F -DVX_RING_PRIO=$p vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/pr$p.so &
done
# cache policy of the LDS-DMA copies (all GEMM kernels): tools/exp_ring_missue.sh <tag> "mi0 pol_nt pol_sc1 pol_sc0sc1"
/opt/rocm/bin/hipcc $F '-DVX_GLDS_MOD=" nt"' vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/pol_nt.so &
/opt/rocm/bin/hipcc $F '-DVX_GLDS_MOD=" sc1"' vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/pol_sc1.so &
/opt/rocm/bin/hipcc $F '-DVX_GLDS_MOD=" sc0 sc1"' vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/pol_sc0sc1.so &
wait
/opt/rocm/bin/hipcc $F -DVX_RING_TRACE vx_gemm.hip vx_gemm_ring.hip vx_norm.hip -x hip vx_api.cpp -o ../../tools/ringlibs/trace.so &
wait
ls -la ../../tools/ringlibs
