#!/bin/bash
# round 3, GPU call F: final validation of HEAD (pipeline refactor for the mixed schedule included): whole GPU suite,
# smoke, default bench line, the other configurations on one GPU, the 8-rank config-4 control flow folded onto this GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03f}
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert|FAILED" | tail -150 > $OUT/${T}_pytest_gpu_summary.log
tail -3 $OUT/${T}_pytest_gpu_summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.log 2>&1; tail -1 $OUT/${T}_smoke.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; tail -c 400 $OUT/${T}_bench.json
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'fps', round(d['value'],3), 'ms', round(d['ms_per_step'],1), d['config'].get('parallelism',''), d.get('same_clip_1gpu_fps',''), d.get('note',''))"; }
timeout 600 python bench.py --frames 64 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_cfg.err | tee $OUT/${T}_bench_F64_1gpu.json | pr F64
timeout 600 python bench.py --frames 124 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_cfg.err | tee $OUT/${T}_bench_F124_1gpu.json | pr F124
timeout 600 python bench.py --size 768 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_cfg.err | tee $OUT/${T}_bench_768_bf16.json | pr 768bf16
timeout 600 python bench.py --size 768 --fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>> $OUT/${T}_cfg.err | tee $OUT/${T}_bench_768_fp8.json | pr 768fp8
VX_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 900 python bench.py --gpus 8 --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline --no-same-clip-1gpu 2>> $OUT/${T}_cfg.err | grep "^{" | tee $OUT/${T}_bench_8rank_folded_mixed.json | pr 8rank_folded
