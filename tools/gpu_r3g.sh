#!/bin/bash
# round 3, GPU call G: the round's total on ONE box - round-2 HEAD (2aec0e9, materialised in tools/r02_tree) against this
# HEAD, interleaved, default bench configuration (BASELINE configs[1]).  The tree is not kept: recreate it with
#   mkdir -p tools/r02_tree && git archive 2aec0e9 | tar -x -C tools/r02_tree --exclude=tests/golden --exclude=profiles
#   make -C tools/r02_tree/v-express_amd/csrc -j8
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
T=${1:-r03g}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
: > $OUT/${T}_round_ab.txt
for rep in 1 2 3; do
  for arm in r02 r03; do
    D=$R; [ "$arm" = "r02" ] && D=$R/tools/r02_tree
    ( cd $D && timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err ) | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_round_ab.txt
  done
done
cat $OUT/${T}_round_ab.txt
