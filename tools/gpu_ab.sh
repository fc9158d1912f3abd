#!/bin/bash
# tests + A/B bench of the ring kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
VX_GEMM_RING=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --gemm-shapes gpurun_out/shapes_ring0.txt > gpurun_out/bench_ring0.log 2> gpurun_out/bench_ring0.err
VX_GEMM_RING=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --gemm-shapes gpurun_out/shapes_ring1.txt > gpurun_out/bench_ring1.log 2> gpurun_out/bench_ring1.err
VX_GEMM_RING=2 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_ring2.log 2> gpurun_out/bench_ring2.err
tail -3 gpurun_out/pytest_gpu.log
for i in 0 1 2; do python -c "import json,sys; d=json.loads(open('gpurun_out/bench_ring$i.log').read().strip().splitlines()[-1]); print('ring=$i', d['value'], d['ms_per_step'], d.get('roofline',{}).get('all_gemm_tflops'))"; done
