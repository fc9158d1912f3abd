#!/bin/bash
# round 4, GPU call E: fused feed-forward schedule variants + skeleton ablations (eight-wave form) + whole-pipeline A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
T=r04e
{
for lib in v-express_amd/libvexpress_hip.so tools/fflibs/pf3.so tools/fflibs/pf4.so tools/fflibs/nosb.so tools/fflibs/pf3nosb.so v-express_amd/libvexpress_hip.so \
           tools/fflibs/abl63.so tools/fflibs/abl127.so tools/fflibs/abl191.so tools/fflibs/abl255.so; do
  echo "=== $lib"
  timeout 100 tools/ff_bench $lib 20 | tail -3
done
} > $OUT/${T}_ff_variants.txt 2>&1
cat $OUT/${T}_ff_variants.txt
bench1() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_ff_fused.txt
}
for rep in 1 2; do
  bench1 "base rep$rep" VX_NOOP=1
  bench1 "VX_FF_FUSED=1 rep$rep" VX_FF_FUSED=1
done
cat $OUT/${T}_ab_ff_fused.txt
