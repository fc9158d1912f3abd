#!/bin/bash
# round 3, GPU call D: STORE epilogue without bias registers, residual prefetch depth 6 / 10 / 12 (kernel-level and
# whole-path A/B), parity of the re-templated ring kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03d}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
NEW=v-express_amd/libvexpress_hip.so
: > $OUT/${T}_res_depth.txt
for rep in 1 2; do
  for lib in tools/ringlibs/prev.so tools/ringlibs/res6.so $NEW tools/ringlibs/res12.so; do
    echo "=== rep $rep lib=$lib" >> $OUT/${T}_res_depth.txt
    for f in "lin" "ffout" "prepad"; do
      timeout 120 tools/gemm_bench $lib 20 "$f" 2>&1 | grep -E "^(L[0-3]|VAE) " | cut -c1-100 >> $OUT/${T}_res_depth.txt
    done
  done
done
python3 - $OUT/${T}_res_depth.txt <<'PY'
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
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or conv or geglu or row_stats or groupnorm_folded or grouped" 2>&1 | tail -12 > $OUT/${T}_kernels.log
tail -3 $OUT/${T}_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "not bench_two_rank" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -40 > $OUT/${T}_models.log
tail -3 $OUT/${T}_models.log
for rep in 1 2; do
  for arm in res6 new res12; do
    L=$NEW; [ "$arm" != "new" ] && L=tools/ringlibs/$arm.so
    VX_LIBRARY=$PWD/$L timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_ab.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$arm fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab.txt
  done
done
cat $OUT/${T}_ab.txt
