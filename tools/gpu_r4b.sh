#!/bin/bash
# round 4, GPU call B: the fused feed-forward prototype (microbenchmark + parity), the batch-invariance fix of the GroupNorm
# partial sums (F28 merged call vs per-window calls), same-box A/B of the FF variants.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=r04b
L=v-express_amd/libvexpress_hip.so
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 120 tools/ff_bench $L 20 > $OUT/${T}_ff_fused_bench.txt 2>&1
cat $OUT/${T}_ff_fused_bench.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -s -k "ff_fused or gn_partial or groupnorm_folded" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -30 > $OUT/${T}_kernel_tests.log
cat $OUT/${T}_kernel_tests.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s -k "two_overlapping or 16_frame_forward" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -30 > $OUT/${T}_fullsize_tests.log
cat $OUT/${T}_fullsize_tests.log
bench1() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_ff.txt
}
fused_ok=$(python - <<'PY'
import re
t = open("gpurun_out/r04b_ff_fused_bench.txt").read()
m = re.findall(r"two launches ([0-9.]+) us .*fused ([0-9.]+) us", t)
ok = "ok" in t.splitlines()[0] if t else False
print(1 if (m and ok and float(m[-1][1]) < float(m[-1][0])) else 0)
PY
)
for rep in 1 2; do
  bench1 "base rep$rep" VX_NOOP=1
  bench1 "VX_FF_SLAB_MB=200 rep$rep" VX_FF_SLAB_MB=200
  if [ "$fused_ok" = "1" ]; then bench1 "VX_FF_FUSED=1 rep$rep" VX_FF_FUSED=1; fi
done
cat $OUT/${T}_ab_ff.txt
if [ "$fused_ok" = "1" ]; then
  VX_FF_FUSED=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s -k "16_frame_forward or 25_step_call" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -20 > $OUT/${T}_fullsize_ff_fused.log
  cat $OUT/${T}_fullsize_ff_fused.log
fi
