cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
T=r05b
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -p no:cacheprovider -k "conv3" > gpurun_out/${T}_conv3_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_conv3_tests.log
grep -E "^\[|passed|failed|rror|assert|pytest exit" gpurun_out/${T}_conv3_tests.log | tail -40
timeout 300 python tools/conv3_bench.py 20 > gpurun_out/${T}_conv3_bench.txt 2>&1
cat gpurun_out/${T}_conv3_bench.txt | tail -12
if grep -q "pytest exit 0" gpurun_out/${T}_conv3_tests.log; then
  bash tools/gpu_job.sh tests ${T}f -k "16_frame_forward or two_overlapping or 25_step_call" tests/test_gpu_fullsize.py
  bash tools/gpu_job.sh ab $T VX_CONV3_GN 0 1 2
fi
