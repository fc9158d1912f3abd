cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
VX_LIBRARY=$PWD/tools/c3libs/trace.so timeout 300 python tools/conv3_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05h_conv3_trace_kkmajor.txt
head -54 gpurun_out/r05h_conv3_trace_kkmajor.txt
timeout 300 python tools/conv3_bench.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r05h_conv3_bench.txt; cat gpurun_out/r05h_conv3_bench.txt
