#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02i}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "folded or row_stats or gemm or conv" 2>&1 | tail -8 > $OUT/${T}_gemm_tests.log
for fold in 1 0 1 0; do
  VX_LN_FOLD=$fold timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline >> $OUT/${T}_bench_fold$fold.json 2>> $OUT/${T}_bench_fold$fold.err
done
bash tools/gpu_profile.sh $T
tail -3 $OUT/${T}_gemm_tests.log
for f in $OUT/${T}_bench_fold1.json $OUT/${T}_bench_fold0.json; do python - "$f" <<'PY'
import json, sys
for ln in open(sys.argv[1]).read().strip().splitlines():
    try:
        d = json.loads(ln); print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
    except Exception as e:
        pass
PY
done
head -12 $OUT/prof_${T}_trace_summary.txt
