#!/bin/bash
# the other single-GPU configurations of BASELINE.json at HEAD: 64 / 124 frames (sliding windows, merged calls), 768x768
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=r04x
for cfg in "F64:--frames 64" "F124:--frames 124" "768:--size 768" "768_fp8:--size 768 --fp8"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $flags > gpurun_out/${T}_bench_${name}.json 2>> gpurun_out/${T}_bench.err
done
