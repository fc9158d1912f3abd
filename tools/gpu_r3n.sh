#!/bin/bash
# round 3, GPU call N: last sanity check of the final HEAD after the context-scheduler rewrite (host-only change)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03n}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "pipeline_vs_reference or merged_unet_calls" 2>&1 | grep -E "^\[|passed|failed|Error|error" > $OUT/${T}_tests.log
cat $OUT/${T}_tests.log
timeout 200 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" | tee $OUT/${T}_bench.txt
