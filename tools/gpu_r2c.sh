#!/bin/bash
# Round-2 GPU call C: whole GPU suite, smoke, bench (N=1 with cpu baseline), rocprofv3 kernel trace, PMC traffic of the
# dominant kernel, and the config-4 control flow with 8 ranks folded onto this one GPU (gloo staging instead of RCCL).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02c}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -120 > $OUT/${T}_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --gemm-shapes $OUT/${T}_gemm_by_shape.txt > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err
bash tools/gpu_profile.sh $T
bash tools/exp_pmc_bench.sh $T > $OUT/${T}_pmc_bench.log 2>&1
VX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 0 --ddim-steps 2 --no-roofline > $OUT/${T}_bench_2rank_folded.json 2> $OUT/${T}_bench_2rank_folded.err
timeout 600 python bench.py --steps 1 --warmup 1 --frames 124 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_F124_1gpu.json 2> $OUT/${T}_bench_F124_1gpu.err
timeout 600 python bench.py --steps 1 --warmup 1 --frames 64 --no-cpu-baseline --no-roofline > $OUT/${T}_bench_F64_1gpu.json 2> $OUT/${T}_bench_F64_1gpu.err
tail -4 $OUT/${T}_pytest_gpu.log; tail -1 $OUT/${T}_smoke.log
for f in bench bench_8rank_folded bench_F124_1gpu bench_F64_1gpu; do python - "$OUT/${T}_$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1), d.get("config", {}).get("workload", "")[:50], d.get("same_clip_1gpu_fps"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -3 $OUT/${T}_bench_2rank_folded.err
