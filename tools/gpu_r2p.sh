#!/bin/bash
# final validation of the round's HEAD: the whole GPU suite, smoke(), one default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r02p}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 530 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider -s 2>&1 | grep -vE "^\s*$" | tail -80 > $OUT/${T}_pytest_gpu_summary.log
tail -3 $OUT/${T}_pytest_gpu_summary.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1; tail -2 $OUT/${T}_smoke.log
timeout 200 python bench.py --no-cpu-baseline > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; tail -c 600 $OUT/${T}_bench.json
