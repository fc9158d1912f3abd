#!/bin/bash
# Full GPU-box round the way the driver does it: pytest -m gpu, smoke, bench, then a rocprofv3 kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r01}
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -30 > $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --gemm-shapes $OUT/gemm_by_shape.txt ${BENCH_ARGS} > $OUT/bench.log 2> $OUT/bench.err
if [ "$2" != "noprof" ]; then bash tools/gpu_profile.sh $TAG; fi
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/bench.log
