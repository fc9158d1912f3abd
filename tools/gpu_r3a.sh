#!/bin/bash
# round 3, GPU call A: measurements that decide the round's kernel work + the new parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
T=${1:-r03a}
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc; } > $OUT/${T}_box.log 2>&1
make -C v-express_amd/csrc -j 2>&1 | tail -2 > $OUT/${T}_build.log
# 1. power / clock around 2-3 s loops of the d = 40 self-attention launch (attn2, attn3, attn3 variant 4)
for v in 0 1 4; do
  VX_ATTN3=$v ATTN_BOUND=1 ATTN_PRESCALED=1 timeout 120 python tools/power_sample.py --hz 25 --tag attn3_variant_$v -- \
    tools/attn_bench v-express_amd/libvexpress_hip.so 2500 "L0 self" > $OUT/${T}_power_attn_v$v.txt 2>&1
done
# the long-K ring conv and the K = 320 GEGLU as comparison points (same sampler)
timeout 120 python tools/power_sample.py --hz 25 --tag ring_conv_L0 -- tools/gemm_bench tools/ringlibs/mi0.so 8000 "L0 conv3x3 320>320 prepad" > $OUT/${T}_power_ringconv.txt 2>&1
timeout 120 python tools/power_sample.py --hz 25 --tag ring_geglu_L0 -- tools/gemm_bench tools/ringlibs/mi0.so 8000 "L0 geglu" > $OUT/${T}_power_geglu.txt 2>&1
# 2. LDS-DMA cache policy A/B (GEMM-only, interleaved)
bash tools/exp_ring_missue.sh ${T}_pol "mi0 pol_nt pol_sc1 pol_sc0sc1" > $OUT/${T}_pol_summary.txt 2>&1
# 3. hipBLASLt yardstick
timeout 600 python tools/blaslt_yardstick.py 20 > $OUT/${T}_blaslt_yardstick.txt 2> $OUT/${T}_blaslt_yardstick.err
# 4. new parity tests
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -40 > $OUT/${T}_fullsize.log
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s -k "full_size or zero_audio or fp8 or reduce" 2>&1 | grep -E "^\[|passed|failed|Error|error|assert" | tail -30 > $OUT/${T}_models_subset.log
timeout 600 python -m pytest tests/test_gpu_checkpoints.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5 > $OUT/${T}_ckpt.log
# 5. bench at this HEAD (short)
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err
tail -3 $OUT/${T}_fullsize.log $OUT/${T}_models_subset.log $OUT/${T}_ckpt.log; cat $OUT/${T}_pol_summary.txt | tail -12; head -30 $OUT/${T}_blaslt_yardstick.txt
for f in $OUT/${T}_power_*.txt; do tail -1 $f | cut -c1-700; done
tail -2 $OUT/${T}_bench.json | cut -c1-400
