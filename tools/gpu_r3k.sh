#!/bin/bash
# round 3, GPU call K: validation of the final HEAD: whole GPU suite, smoke, the driver's bench command shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03k}
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -150 > $OUT/${T}_pytest_gpu_summary.log
tail -3 $OUT/${T}_pytest_gpu_summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1; tail -1 $OUT/${T}_smoke.log
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; tail -c 300 $OUT/${T}_bench.json
