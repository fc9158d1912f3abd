#!/bin/bash
# Round-2 GPU call G: LayerNorm folded into the consumer GEMMs - kernel tests, model parity, A/B bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02g}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "folded or row_stats" 2>&1 | tail -30 > $OUT/${T}_fold_tests.log
timeout 1800 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -60 > $OUT/${T}_model_tests.log
for fold in 1 0 1 0; do
  VX_LN_FOLD=$fold timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline >> $OUT/${T}_bench_fold$fold.json 2>> $OUT/${T}_bench_fold$fold.err
done
tail -6 $OUT/${T}_fold_tests.log; grep -E "passed|failed|FAILED|fullsize" $OUT/${T}_model_tests.log | tail -12
for f in $OUT/${T}_bench_fold1.json $OUT/${T}_bench_fold0.json; do python - "$f" <<'PY'
import json, sys
for ln in open(sys.argv[1]).read().strip().splitlines():
    try:
        d = json.loads(ln); print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
    except Exception as e:
        pass
PY
done
