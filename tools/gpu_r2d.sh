#!/bin/bash
# Round-2 GPU call D: fp8 on the ring kernel (numerics + 768^2 A/B), split-K policy test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02d}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "fp8 or split_k" 2>&1 | grep -v "^$" | tail -40 > $OUT/${T}_fp8_tests.log
for ring in 1 0; do
  VX_FP8_RING=$ring timeout 900 python bench.py --steps 1 --warmup 1 --size 768 --fp8 --no-cpu-baseline --gemm-shapes $OUT/${T}_shapes_768_fp8_ring$ring.txt > $OUT/${T}_bench_768_fp8_ring$ring.json 2> $OUT/${T}_bench_768_fp8_ring$ring.err
  VX_FP8_RING=$ring timeout 900 python bench.py --steps 2 --warmup 1 --fp8 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_512_fp8_ring$ring.json 2> $OUT/${T}_bench_512_fp8_ring$ring.err
done
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_512_bf16.json 2> $OUT/${T}_bench_512_bf16.err
timeout 900 python bench.py --steps 1 --warmup 1 --size 768 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_768_bf16.json 2> $OUT/${T}_bench_768_bf16.err
tail -5 $OUT/${T}_fp8_tests.log
for f in $OUT/${T}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep fp8 $OUT/${T}_shapes_768_fp8_ring1.txt | head -12
