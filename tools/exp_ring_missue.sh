#!/bin/bash
# A/B of the experimental ring-kernel issue placements (VX_RING_MISSUE 0-3 -> mi*.so, VX_RING_PRIO 0/2 -> pr*.so, built by
# tools/build_ring_variants.sh): GEMM-only, interleaved so that box drift hits every arm alike
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-r02q}
ARMS=${2:-"mi0 mi1 mi2"}
O=gpurun_out/${T}_ring_missue.txt
: > $O
for rep in 1 2; do
  for m in $ARMS; do
    echo "=== rep $rep arm=$m" >> $O
    for f in "prepad" "ffout" "L0 lin"; do
      timeout 40 tools/gemm_bench tools/ringlibs/$m.so 20 "$f" 2>&1 | grep -E "^(L|VAE)" | cut -c1-110 >> $O
    done
  done
done
for m in $ARMS; do
  [ -f tools/ringlibs/${m}_trace.so ] || continue
  echo "=== trace arm=$m" >> $O
  RING_TRACE=1 timeout 40 tools/gemm_bench tools/ringlibs/${m}_trace.so 3 "L0 conv3x3 320>320 prepad" 2>&1 | head -16 >> $O
done
python3 - $O <<'PY'
import sys, collections, re
t = collections.defaultdict(lambda: collections.defaultdict(list)); arm = None
for ln in open(sys.argv[1]):
    m = re.match(r"=== rep \d+ arm=(\S+)", ln)
    if m: arm = m.group(1); continue
    if ln.startswith("=== trace"): break
    p = ln.split()
    try:
        i = [k for k, x in enumerate(p) if x.isdigit()][0]
        name = " ".join(p[:i]); us = float(p[i + 3]); ok = p[-1]
        t[name][arm].append((us, ok))
    except Exception:
        pass
for name, d in t.items():
    print(f"{name:34s}", "  ".join(f"{a}: {min(u for u, _ in v):7.1f} us {'/'.join(sorted({o for _, o in v}))}" for a, v in sorted(d.items())))
PY
