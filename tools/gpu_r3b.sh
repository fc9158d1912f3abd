#!/bin/bash
# round 3, GPU call B: parity of the producer-side statistics / GroupNorm fold, whole GPU suite, same-box A/B of the
# two fusions, power sampling on the right card, kernel trace of the bench at this HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03b}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "row_stats or groupnorm_folded or grouped_weights or ring or folded_layernorm or fp8" 2>&1 | tail -40 > $OUT/${T}_kernels_new.log
tail -3 $OUT/${T}_kernels_new.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|^tests/|FAILED" | tail -120 > $OUT/${T}_pytest_gpu_summary.log
tail -4 $OUT/${T}_pytest_gpu_summary.log
for rep in 1 2; do
  for arm in "0 0" "1 1" "1 0" "0 1"; do
    set -- $arm
    VX_GN_FOLD=$1 VX_FUSED_STATS=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gnfold=$1 fusedstats=$2 fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab.txt
  done
done
cat $OUT/${T}_ab.txt
for v in 0 1 4; do
  VX_ATTN3=$v ATTN_BOUND=1 ATTN_PRESCALED=1 timeout 120 python tools/power_sample.py --hz 25 --tag attn3_variant_$v -- \
    tools/attn_bench v-express_amd/libvexpress_hip.so 2500 "L0 self" > $OUT/${T}_power_attn_v$v.txt 2>&1
  tail -1 $OUT/${T}_power_attn_v$v.txt | cut -c1-900
done
timeout 120 python tools/power_sample.py --hz 25 --tag ring_conv_L0 -- tools/gemm_bench v-express_amd/libvexpress_hip.so 8000 "L0 conv3x3 320>320 prepad" > $OUT/${T}_power_ringconv.txt 2>&1
tail -1 $OUT/${T}_power_ringconv.txt | cut -c1-900
bash tools/gpu_profile.sh $T > /dev/null 2>&1
head -45 $OUT/prof_${T}_trace_summary.txt
tail -c 1500 $OUT/prof_${T}_bench.log
