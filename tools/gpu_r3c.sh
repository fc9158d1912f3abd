#!/bin/bash
# round 3, GPU call C: GEGLU epilogue (merged bias / LayerNorm FMA, tail-polynomial GELU) and 8-heads-per-block temporal
# attention: kernel-level A/B, parity, same-box whole-path A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03c}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
NEW=v-express_amd/libvexpress_hip.so
: > $OUT/${T}_geglu_ab.txt
for rep in 1 2; do
  for ln in 0 1; do
    for lib in tools/ringlibs/prev.so tools/ringlibs/gelu_as.so $NEW; do
      echo "=== rep $rep LNFOLD=$ln lib=$lib" >> $OUT/${T}_geglu_ab.txt
      LNFOLD=$ln timeout 120 tools/gemm_bench $lib 20 "geglu" 2>&1 | grep -E "^L[0-3] " | cut -c1-100 >> $OUT/${T}_geglu_ab.txt
    done
  done
done
cat $OUT/${T}_geglu_ab.txt | grep -E "===|L0|L1" | paste - - - | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "geglu or gelu or temporal or folded or epilogue or groupnorm or row_stats" 2>&1 | tail -15 > $OUT/${T}_kernels.log
tail -3 $OUT/${T}_kernels.log
timeout 900 python -m pytest tests/test_gpu_prologue.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5 > $OUT/${T}_prologue.log
tail -2 $OUT/${T}_prologue.log
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -60 > $OUT/${T}_models.log
tail -4 $OUT/${T}_models.log
for rep in 1 2; do
  for arm in "new 8" "new 4" "gelu_as 8"; do
    set -- $arm
    L=$NEW; [ "$1" = "gelu_as" ] && L=tools/ringlibs/gelu_as.so
    VX_LIBRARY=$PWD/$L VX_TEMPORAL_WPB=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$1 temporal_wpb=$2 fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab.txt
  done
done
cat $OUT/${T}_ab.txt
