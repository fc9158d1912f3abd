#!/bin/bash
# round 3, GPU call H: Q | K on the ring kernel + V^T on the classic SPLIT tiles - parity subset and same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03h}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "fullsize_16_frame or fullsize_25_step_call_vs or full_size_unet_forward or ring_vs_classic or golden" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -30 > $OUT/${T}_tests.log
tail -4 $OUT/${T}_tests.log
for rep in 1 2 3; do
  for arm in 0 1; do
    VX_QK_RING=$arm timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qk_ring=$arm fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab.txt
  done
done
cat $OUT/${T}_ab.txt
