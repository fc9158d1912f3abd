cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
T=r05i
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -p no:cacheprovider -k "tblock" > gpurun_out/${T}_tblock_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_tblock_tests.log
grep -E "^\[|passed|failed|rror|assert|pytest exit" gpurun_out/${T}_tblock_tests.log | tail -24
{ timeout 200 python tools/tb_bench.py 30 16; timeout 200 python tools/tb_bench.py 30 24; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tb_bench.txt
cat gpurun_out/${T}_tb_bench.txt
