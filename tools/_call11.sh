cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PYTHONPATH
T=r05j
bash tools/gpu_job.sh tests $T -k "default_window_24" tests/test_gpu_fullsize.py
for rep in 1 2; do for v in 0 1; do
  r=$(VX_TB_FUSED=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --frames 44 --context-frames 24 2>> gpurun_out/${T}_ab.err | python -c "
import sys, json
v=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(round(v[-1]['value'],3), round(v[-1]['ms_per_step'],1), [x for k,x in v[-1]['block_paths'].items() if 'temporal_attention c=320' in k])")
  echo "ctx24 F44 VX_TB_FUSED=$v rep $rep: $r" | tee -a gpurun_out/${T}_ab_ctx24_tblock.txt
done; done
