#!/bin/bash
# attn3 (d = 40 attention): compile-time ablations of the loop body + SQ counters, self-attention shape of the 64x64 level.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r02b}
O=$PWD/gpurun_out/${T}_attn3_ablation.txt
: > $O
for v in 1 2 3; do
  for m in 0 1 2 3 4 7 8 16 24 32 63; do
    printf "variant %s mask %-3s " $v $m >> $O
    VX_ATTN3=$v ATTN_BOUND=1 ATTN_PRESCALED=1 timeout 120 tools/attn_bench tools/attnlibs/abl$m.so 10 "L0 self" 2>&1 | grep "^L0 " | cut -c1-100 >> $O
  done
done
for v in 0 1 2 3; do
  printf "variant %s general scale (v_mul path) " $v >> $O
  VX_ATTN3=$v ATTN_BOUND=1 timeout 120 tools/attn_bench v-express_amd/libvexpress_hip.so 10 "L0" 2>&1 | grep "^L0 " | cut -c1-100 >> $O
done
L=$PWD/v-express_amd/libvexpress_hip.so
B=$PWD/tools/attn_bench
P=$PWD/gpurun_out/${T}_attn3_pmc.txt
: > $P
cd /tmp
for v in 0 2 3; do
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pmc_out
    VX_ATTN3=$v ATTN_BOUND=1 ATTN_PRESCALED=1 timeout 180 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_out -o run -- $B $L 3 "L0 self" > /dev/null 2>&1
    f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
    echo "=== VX_ATTN3=$v pmc=[$pmc]" >> $P
    if [ -n "$f" ]; then
      python3 - "$f" >> $P <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:70]
    if "attn" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
    else echo "no counter file" >> $P; fi
  done
done
cat $O
