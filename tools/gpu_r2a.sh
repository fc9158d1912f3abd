#!/bin/bash
# Round-2 GPU call A: attn3 variants (micro-benchmark + numerics), new kernel tests, the whole GPU suite incl. the
# full-size goldens, bench (bf16, fp8 768^2, attn A/B), rocprofv3 kernel trace, folded 2-rank run of the self-launching bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02a}
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; } > $OUT/${T}_box.log 2>&1
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
for v in 0 1 2 3; do
  echo "== VX_ATTN3=$v" >> $OUT/${T}_attn_bench.txt
  VX_ATTN3=$v ATTN_BOUND=1 timeout 300 tools/attn_bench v-express_amd/libvexpress_hip.so 10 "L0" >> $OUT/${T}_attn_bench.txt 2>&1
done
for v in 1 2 3; do
  echo "== VX_ATTN3=$v" >> $OUT/${T}_attn_tests.log
  VX_ATTN3=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or bounded" 2>&1 | tail -25 >> $OUT/${T}_attn_tests.log
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "fp8 or layout_and_loop" 2>&1 | tail -40 > $OUT/${T}_fp8_tests.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -150 > $OUT/${T}_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --gemm-shapes $OUT/${T}_gemm_by_shape.txt > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err
bash tools/gpu_profile.sh $T
VX_ATTN3=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_attn2.json 2> $OUT/${T}_bench_attn2.err
VX_ATTN3=3 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_attn3pipe.json 2> $OUT/${T}_bench_attn3pipe.err
timeout 900 python bench.py --steps 1 --warmup 1 --size 768 --fp8 --no-cpu-baseline --gemm-shapes $OUT/${T}_gemm_by_shape_768_fp8.txt > $OUT/${T}_bench_768_fp8.json 2> $OUT/${T}_bench_768_fp8.err
timeout 900 python bench.py --steps 1 --warmup 1 --size 768 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_768_bf16.json 2> $OUT/${T}_bench_768_bf16.err
VX_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --frames 28 --steps 1 --warmup 0 --ddim-steps 2 --no-roofline > $OUT/${T}_bench_2rank_folded.json 2> $OUT/${T}_bench_2rank_folded.err
tail -4 $OUT/${T}_pytest_gpu.log; cat $OUT/${T}_attn_bench.txt | grep -E "==|L0"; tail -1 $OUT/${T}_smoke.log
for f in bench bench_attn2 bench_attn3pipe bench_768_fp8 bench_768_bf16 bench_2rank_folded; do python - "$OUT/${T}_$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1), d.get("config", {}).get("workload", "")[:60])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
