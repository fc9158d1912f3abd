#!/bin/bash
# round 4, GPU call A: new GroupNorm-partial-sum epilogues (kernel tests), the F28 / 768 reference goldens at the benchmarked
# geometry, microbenchmarks (packed-FMA issue rate, packed GELU A/B, 128x320 tile and forced ring on the 16x16 level, ring
# ablation 24), same-box A/B of the fused GroupNorm statistics, one bench line with the new roofline fields.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=r04a
L=v-express_amd/libvexpress_hip.so
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; } > $OUT/${T}_box.log 2>&1
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 60 tools/valu_bench > $OUT/${T}_valu_issue_rates.txt 2>&1
{
echo "# L2 (16x16-level) shapes: product tiles | VX_GEMM_T128X320=1 | RING_HINT=1 (persistent 256x320 kernel on 128 CUs)"
for rep in 1 2; do
  echo "=== rep $rep default";      timeout 120 tools/gemm_bench $L 20 "L2 " | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
  echo "=== rep $rep T128X320";     VX_GEMM_T128X320=1 timeout 120 tools/gemm_bench $L 20 "L2 " | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
  echo "=== rep $rep RING_HINT=1";  RING_HINT=1 timeout 120 tools/gemm_bench $L 20 "L2 " | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
done
} > $OUT/${T}_l2_tiles.txt 2>&1
{
echo "# GEGLU epilogue: product (six v_fmaak_f32 per element) vs VX_GELU_PK (six v_pk_fma_f32 per element pair)"
for rep in 1 2 3; do
  for lib in $L tools/ringlibs/gelu_pk.so; do
    for ln in 0 1; do
      echo "=== rep $rep LNFOLD=$ln lib=$lib"; LNFOLD=$ln timeout 120 tools/gemm_bench $lib 20 "geglu" | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
    done
  done
done
} > $OUT/${T}_gelu_pk_ab.txt 2>&1
{
echo "# ring ablation: mask 24 = MFMAs + LDS fragment reads, NO operand copies (pure compute skeleton) vs product"
for lib in $L tools/ringlibs/abl24.so; do
  echo "=== lib=$lib"; timeout 120 tools/gemm_bench $lib 20 "prepad" | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
  timeout 60 tools/gemm_bench $lib 20 "L0 ffout" | grep -v "^norm\|^L. [0-9]\|^VAE\|^shape\|^weighted"
done
} > $OUT/${T}_ring_abl24.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gn_partial or groupnorm or row_stats or split_k or epilogue_options" 2>&1 | tail -40 > $OUT/${T}_kernel_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -60 > $OUT/${T}_fullsize_tests.log
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "zero_audio or merged or window_geometry or pipeline_vs_reference or forward" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -40 > $OUT/${T}_model_tests.log
for rep in 1 2; do
  for gn in 0 1; do
    VX_GN_FUSED=$gn timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VX_GN_FUSED=$gn rep $rep fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_gn_fused.txt
  done
done
timeout 400 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --gemm-shapes $OUT/${T}_gemm_by_shape.txt > $OUT/${T}_bench.json 2>> $OUT/${T}_bench.err
cat $OUT/${T}_ab_gn_fused.txt; tail -3 $OUT/${T}_kernel_tests.log; tail -3 $OUT/${T}_fullsize_tests.log; tail -3 $OUT/${T}_model_tests.log
