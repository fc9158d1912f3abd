#!/bin/bash
# round 3, GPU call M: one problem, three tilings - operand bytes per clock through the CU's L1 path
#   ring 256x320 (6.9 KB/MFLOP) | classic 256x320, one 8-wave workgroup per CU (6.9) | classic 128x160, two 4-wave workgroups (14.1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r03m}_tilings.txt
L=v-express_amd/libvexpress_hip.so
: > $OUT
for rep in 1 2; do
  for mode in "ring:VX_GEMM_RING=2" "classic256x320:VX_GEMM_RING=0" "classic128x160:VX_GEMM_RING=0 VX_GEMM_TILE=small"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo "=== rep $rep $name" >> $OUT
    for f in "L0 conv3x3 320>320 prepad" "L1 conv3x3 640>640 prepad" "L2 conv3x3 1280>1280 prepad" "L0 ffout"; do
      env $envs timeout 60 tools/gemm_bench $L 20 "$f" 2>&1 | grep -E "^L[0-3] " | cut -c1-100 >> $OUT
    done
  done
done
cat $OUT
