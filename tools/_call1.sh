cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05a
bash tools/gpu_job.sh box $T
{ echo "# conv K loop: product (abl0) vs mask 8 = NO A copies (B copies, fragment reads, MFMAs, epilogue stay) - the best case of a halo-resident A operand"
for rep in 1 2; do for l in abl0 abl8; do echo "=== rep $rep $l"; timeout 120 tools/gemm_bench tools/ringlibs/$l.so 20 "prepad" | grep "^L[01] "; done; done; } > gpurun_out/${T}_ring_no_a_copies.txt 2>&1
bash tools/gpu_job.sh tests $T -k "default_window_24 or 16_frame_forward" tests/test_gpu_fullsize.py
bash tools/gpu_job.sh bench $T
bash tools/gpu_job.sh configs $T
cat gpurun_out/${T}_ring_no_a_copies.txt | tail -20
