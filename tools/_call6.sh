cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
VX_LIBRARY=$PWD/tools/c3libs/trace.so timeout 300 python tools/conv3_trace.py > gpurun_out/r05f_conv3_trace.txt 2>&1
cat gpurun_out/r05f_conv3_trace.txt | grep -v amdgpu.ids | head -120
