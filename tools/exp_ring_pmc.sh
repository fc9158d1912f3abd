#!/bin/bash
# where does the long-K ring GEMM stall?  LDS / vector-memory-path counters of one conv shape (separate PMC passes;
# unknown counter names just fail their own pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r02s}
L=$PWD/${2:-v-express_amd/libvexpress_hip.so}
B=$PWD/tools/gemm_bench
SHAPE=${3:-L0 conv3x3 320>320 prepad}
O=$PWD/gpurun_out/${T}_ring_pmc.txt
: > $O
cd /tmp
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
  rm -rf /tmp/pmc_out
  timeout 60 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_out -o run -- $B $L 3 "$SHAPE" > /tmp/gb.log 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  echo "=== pmc=[$pmc]" >> $O
  if [ -n "$f" ]; then
    python3 - "$f" >> $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:48]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
  else
    echo "no counter file: $(grep -i -m2 -E "error|invalid|not found|unknown" /tmp/gb.log | cut -c1-200)" >> $O
  fi
done
cat $O | cut -c1-400
