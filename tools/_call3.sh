cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
T=${1:-r05c}
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -p no:cacheprovider -k "conv3" > gpurun_out/${T}_conv3_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_conv3_tests.log
grep -E "^\[|passed|failed|rror|assert|pytest exit" gpurun_out/${T}_conv3_tests.log | tail -14
{ echo "=== product"; timeout 300 python tools/conv3_bench.py 20 2>&1 | grep -v amdgpu.ids
  for l in $2; do echo "=== $l"; VX_LIBRARY=$PWD/tools/c3libs/$l.so timeout 300 python tools/conv3_bench.py 20 2>&1 | grep "^L[01]"; done; } > gpurun_out/${T}_conv3_bench.txt
cat gpurun_out/${T}_conv3_bench.txt
if grep -q "pytest exit 0" gpurun_out/${T}_conv3_tests.log; then
  bash tools/gpu_job.sh ab $T VX_CONV3_GN 0 1 2
fi
