#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02k}
G=$OUT/${T}_gemm_stages.txt
: > $G
for st in 2 3 4; do
  echo "== VX_GEMM_STAGES128=$st" >> $G
  VX_GEMM_STAGES128=$st timeout 200 tools/gemm_bench v-express_amd/libvexpress_hip.so 10 "L2 " 2>&1 | grep "^L2" | cut -c1-110 >> $G
  VX_GEMM_STAGES128=$st timeout 200 tools/gemm_bench v-express_amd/libvexpress_hip.so 10 "L3 lin" 2>&1 | grep "^L3" | cut -c1-110 >> $G
done
for st in 2 3 4 2 3 4; do
  VX_GEMM_STAGES128=$st timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline >> $OUT/${T}_bench_st$st.json 2>> $OUT/${T}_bench_st$st.err
done
cat $G
for st in 2 3 4; do python - "$OUT/${T}_bench_st$st.json" <<'PY'
import json, sys
for ln in open(sys.argv[1]).read().strip().splitlines():
    try:
        d = json.loads(ln); print(sys.argv[1], "fps", round(d["value"], 3), "ms", round(d["ms_per_step"], 1))
    except Exception as e:
        pass
PY
done
