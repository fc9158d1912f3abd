#!/bin/bash
# round 4, GPU call F: 256x256 tile on the VAE convolutions (gemm_bench A/B + decode parity), then the whole GPU suite at HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=r04f
L=v-express_amd/libvexpress_hip.so
make -C v-express_amd/csrc -j 2>&1 | tail -1 > $OUT/${T}_build.log
{
for rep in 1 2; do
  echo "=== rep $rep VX_GEMM_T256X256=0 (128x128 tile)"; VX_GEMM_T256X256=0 timeout 200 tools/gemm_bench $L 10 "VAE" | grep -v "^norm\|^L. [0-9]\|^shape\|^weighted\|^VAE 512^2 128 x4"
  echo "=== rep $rep 256x256 tile";                    timeout 200 tools/gemm_bench $L 10 "VAE" | grep -v "^norm\|^L. [0-9]\|^shape\|^weighted\|^VAE 512^2 128 x4"
done
} > $OUT/${T}_vae_tile_256x256.txt 2>&1
cat $OUT/${T}_vae_tile_256x256.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -120 > $OUT/${T}_pytest_gpu_summary.log
tail -25 $OUT/${T}_pytest_gpu_summary.log
for rep in 1 2; do
  for v in 0 1; do
    VX_GEMM_T256X256=$v timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_bench.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VX_GEMM_T256X256=$v rep $rep fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1))" >> $OUT/${T}_ab_vae_tile.txt
  done
done
cat $OUT/${T}_ab_vae_tile.txt
