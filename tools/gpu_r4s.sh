#!/bin/bash
# multi-rank control flow at HEAD on ONE GPU: 8 ranks folded onto it over gloo (BASELINE configs[3]: 124 frames, mixed schedule
# = whole units + frame-sharded left-overs; the fused temporal block then runs on the pixel-shard layout).  Not a measurement.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=r04s
VX_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 900 python bench.py --gpus 8 --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline --no-same-clip-1gpu 2> gpurun_out/${T}_8rank.err | grep "^{" > gpurun_out/${T}_bench_8rank_folded.json
echo "exit ${PIPESTATUS[0]}" >> gpurun_out/${T}_8rank.err
tail -5 gpurun_out/${T}_8rank.err
