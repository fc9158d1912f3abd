#!/bin/bash
# round 4, GPU call I: XCD-contiguous row ownership in the ring / FF / GroupNorm kernels (VX_XCD_ROWS): kernel parity + whole-path A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=r04i
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "ring or gn_partial or row_stats or ff_fused or groupnorm or geglu or conv" 2>&1 | tail -8 > $OUT/${T}_kernel_tests.log
cat $OUT/${T}_kernel_tests.log
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then libenv="VX_NOOP=1"; else libenv="VX_LIBRARY=$PWD/tools/ringlibs/xcd_rows0.so"; fi
    env $libenv timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VX_XCD_ROWS=$v rep $rep fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_xcd_rows.txt
  done
done
cat $OUT/${T}_ab_xcd_rows.txt
