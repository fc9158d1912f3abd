#!/bin/bash
# round 3, GPU call L: re-check after the lgkmcnt(0) in front of the STATS epilogue's barrier
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03l}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "row_stats_out or groupnorm_folded or ring_linear or ring_store" 2>&1 | tail -4 > $OUT/${T}_kernels.log
tail -2 $OUT/${T}_kernels.log
timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s -k "fullsize_16_frame" 2>&1 | grep -E "^\[|passed|failed" > $OUT/${T}_fullsize.log
cat $OUT/${T}_fullsize.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" | tee $OUT/${T}_bench.txt
