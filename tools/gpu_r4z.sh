#!/bin/bash
# round 4 final: whole GPU suite, smoke, bench with roofline, rocprofv3 trace, PMC traffic - all with the frozen kernel sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
T=${1:-r04z}
rocm-smi --showclocks 2>/dev/null | head -12 > gpurun_out/${T}_box.log
python tools/lib_id.py > gpurun_out/${T}_lib_id.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1
timeout 900 python bench.py --gemm-shapes gpurun_out/${T}_gemm_by_shape.txt > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 900 bash tools/gpu_profile.sh ${T}
timeout 1500 bash tools/exp_pmc_bench.sh ${T}
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench2.json 2>> gpurun_out/${T}_bench.err
