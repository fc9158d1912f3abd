#!/bin/bash
# round 4, GPU call H: cache policy of the ring STORE epilogue's output stores (plain | sc1 | sc0 sc1 | nt), whole-path A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=r04h
for rep in 1 2; do
  for v in plain sc1 sc0sc1 nt; do
    if [ $v = plain ]; then libenv="VX_NOOP=1"; else libenv="VX_LIBRARY=$PWD/tools/ringlibs/store_$v.so"; fi
    env $libenv timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stores=$v rep $rep fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_store_policy.txt
  done
done
cat $OUT/${T}_ab_store_policy.txt
