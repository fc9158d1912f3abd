#!/bin/bash
# Round-2 GPU call B: attention tests after the precision fix, attn3 ablations + SQ counters, small-M GEMM tile / split-K
# policy experiments, fp8 kernel timing vs bf16 per shape.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02b}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
for v in 0 2 3; do
  echo "== VX_ATTN3=$v" >> $OUT/${T}_attn_tests.log
  VX_ATTN3=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or bounded" 2>&1 | tail -25 >> $OUT/${T}_attn_tests.log
done
bash tools/exp_attn3.sh $T > /dev/null 2>&1
# ---- 8x8 / 16x16-level GEMMs: tile and split-K
G=$OUT/${T}_gemm_small.txt
: > $G
for small in 0 1; do for sk in 1 2 4; do
  echo "== VX_GEMM_SMALL64=$small SPLITK=$sk" >> $G
  VX_GEMM_SMALL64=$small SPLITK=$sk timeout 200 tools/gemm_bench v-express_amd/libvexpress_hip.so 10 "L3 " 2>&1 | grep "^L3" | cut -c1-110 >> $G
done; done
echo "== L2 lin (16x16 level), small64 irrelevant" >> $G
timeout 200 tools/gemm_bench v-express_amd/libvexpress_hip.so 10 "L2 " 2>&1 | grep "^L2" | cut -c1-110 >> $G
# ---- whole path A/B
for cfg in "VX_GEMM_SMALL64=0" "VX_GEMM_SMALL64=1" "VX_GEMM_SMALL64=1 VX_SPLITK_ROWS=64" "VX_GEMM_SMALL64=1 VX_SPLITK_ROWS=64 VX_SPLITK_TARGET=256"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-shapes $OUT/${T}_shapes_$tag.txt > $OUT/${T}_bench_$tag.json 2> $OUT/${T}_bench_$tag.err
done
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "two_rank or checkpoint or fp8 or prescaled or batch_invariance" 2>&1 | tail -15 > $OUT/${T}_pytest_subset.log
cat $OUT/${T}_attn3_ablation.txt; cat $G; tail -3 $OUT/${T}_pytest_subset.log; grep -E "passed|failed" $OUT/${T}_attn_tests.log
for f in $OUT/${T}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
