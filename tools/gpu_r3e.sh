#!/bin/bash
# round 3, GPU call E: validation of HEAD (whole GPU suite, smoke, default bench line incl. cpu baseline), kernel trace +
# PMC traffic of the same command, and the classic-tile check of the GELU variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03e}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
: > $OUT/${T}_classic_gelu_variants.txt
for rep in 1 2; do
  for lib in tools/ringlibs/prev.so tools/ringlibs/gelu_as.so tools/ringlibs/gelu_noasm.so v-express_amd/libvexpress_hip.so; do
    echo "=== rep $rep lib=$lib" >> $OUT/${T}_classic_gelu_variants.txt
    for f in "VAE" "L2 conv3x3 1280>1280" "L2 lin" "L3 conv3x3 1280>1280"; do
      timeout 120 tools/gemm_bench $lib 20 "$f" 2>&1 | grep -E "^(L[0-3]|VAE) " | cut -c1-100 >> $OUT/${T}_classic_gelu_variants.txt
    done
  done
done
python3 - $OUT/${T}_classic_gelu_variants.txt <<'PY'
import sys, collections, re
t = collections.defaultdict(lambda: collections.defaultdict(list)); arm = None
for ln in open(sys.argv[1]):
    m = re.match(r"=== rep \d+ lib=(\S+)", ln)
    if m: arm = m.group(1).split("/")[-1]; continue
    p = ln.split()
    try:
        i = [k for k, x in enumerate(p) if x.isdigit()][0]
        t[" ".join(p[:i])][arm].append((float(p[i + 3]), p[-1]))
    except Exception: pass
for name, d in t.items():
    print(f"{name:32s}", "  ".join(f"{a}: {min(u for u, _ in v):7.1f} {'/'.join(sorted({o for _, o in v}))}" for a, v in d.items()))
PY
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -150 > $OUT/${T}_pytest_gpu_summary.log
tail -3 $OUT/${T}_pytest_gpu_summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1; tail -2 $OUT/${T}_smoke.log
timeout 600 python bench.py --gemm-shapes $OUT/${T}_gemm_by_shape.txt > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; tail -c 700 $OUT/${T}_bench.json
bash tools/gpu_profile.sh $T > /dev/null 2>&1
head -30 $OUT/prof_${T}_trace_summary.txt
bash tools/exp_pmc_bench.sh $T > $OUT/${T}_pmc.log 2>&1; head -12 $OUT/pmc_traffic_$T.txt
