#!/bin/bash
# round 3, GPU call I: 64 x 160 tile threshold (VX_GEMM_SMALL64_BELOW = 256 product / 257 / 513), bench A/B + the shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03j}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
for rep in 1 2 3; do
  for arm in 64 16 8 32; do
    VX_GN_APPLY_SLICES=$arm timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_apply_slices=$arm fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab.txt
  done
done
cat $OUT/${T}_ab.txt
