#!/bin/bash
# One GPU-box round: rebuild, per-kernel parity (one pytest process per group so a GPU fault cannot take the
# others down), model parity, smoke, short bench.  Logs go to gpurun_out/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; } > $OUT/box.log 2>&1
make -C v-express_amd/csrc -j 2>&1 | tail -3 > $OUT/build.log
TEST_GROUPS="test_gemm_plain test_gemm_epilogue_options test_conv test_geglu test_gemm_split_qkv_vt test_groupnorm test_layernorm test_flash_attention test_temporal_attention test_small_kv_attention test_add_row_bias test_layout_and_loop_kernels test_errors_are_reported_not_fatal"
: > $OUT/kernels.log
for g in $TEST_GROUPS; do
  echo "=== $g" >> $OUT/kernels.log
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "$g" 2>&1 | tail -40 >> $OUT/kernels.log
done
grep -E "^===|passed|failed|error" $OUT/kernels.log > $OUT/kernels_summary.log
if [ "$1" != "kernels" ]; then
  timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | tail -80 > $OUT/models.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  timeout 900 python bench.py --steps 1 --warmup 1 ${BENCH_ARGS} > $OUT/bench.log 2> $OUT/bench.err
fi
tail -5 $OUT/kernels_summary.log
