#!/bin/bash
# L2 hit rate / fabric traffic of the GEMM kernels on one conv shape (rocprofv3 PMC, separate passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/v-express_amd/libvexpress_hip.so
B=$PWD/tools/gemm_bench
SHAPE=${1:-L0 conv3x3 320>320 prepad}
cd /tmp
for mode in 0 1; do
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    rm -rf /tmp/pmc_out
    VX_GEMM_RING=$mode timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_out -o run -- $B $L 3 "$SHAPE" > /dev/null 2>&1
    f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
    echo "=== ring=$mode pmc=[$pmc]"
    if [ -n "$f" ]; then
      python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: v / cnt[(k, c)] for c, v in d.items()})
PY
    else echo "no counter file"; ls -R /tmp/pmc_out | head; fi
  done
done
