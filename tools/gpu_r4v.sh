#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=r04v
for u in 4 8 4 8 6; do
  echo "VX_UNITS_PER_CALL=$u" >> gpurun_out/${T}_units_per_call.txt
  VX_UNITS_PER_CALL=$u timeout 600 python bench.py --frames 124 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>>gpurun_out/${T}_bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_units_per_call.txt
done
