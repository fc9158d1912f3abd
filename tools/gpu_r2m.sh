#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02m}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
VX_ATTN3=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or bounded" 2>&1 | tail -25 > $OUT/${T}_attn_tests_qk32.log
for v in 1 4; do
  echo "== VX_ATTN3=$v prescaled" >> $OUT/${T}_attn_bench.txt
  VX_ATTN3=$v ATTN_BOUND=1 ATTN_PRESCALED=1 timeout 300 tools/attn_bench v-express_amd/libvexpress_hip.so 10 "L0" >> $OUT/${T}_attn_bench.txt 2>&1
  echo "== VX_ATTN3=$v general" >> $OUT/${T}_attn_bench.txt
  VX_ATTN3=$v ATTN_BOUND=1 timeout 300 tools/attn_bench v-express_amd/libvexpress_hip.so 10 "L0" >> $OUT/${T}_attn_bench.txt 2>&1
done
for v in 1 4 1 4; do
  VX_ATTN3=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline >> $OUT/${T}_bench_attn$v.json 2>> $OUT/${T}_bench_attn$v.err
done
tail -4 $OUT/${T}_attn_tests_qk32.log; grep -E "==|L0" $OUT/${T}_attn_bench.txt
for v in 1 4; do python - "$OUT/${T}_bench_attn$v.json" <<'PY'
import json, sys
for ln in open(sys.argv[1]).read().strip().splitlines():
    try:
        d = json.loads(ln); print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
    except Exception as e:
        pass
PY
done
