#!/bin/bash
# matrix-pipe / LDS / wait counters of the two fused kernels of the 64x64 level (tblock_kernel, ff_fused_kernel) from
# tools/tb_bench.py and one bench step (separate PMC passes with --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export PYTHONPATH=$PWD:$PYTHONPATH
T=${1:-r04w}
R=$PWD
O=$R/gpurun_out/${T}_fused_pmc.txt
echo "# lib_sha256=$(python tools/lib_id.py)  command: python bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-roofline" > $O
cd /tmp
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pmc_out
  ( cd $R && timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_out -o run -- \
      python bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-roofline > /tmp/fp.log 2>&1 )
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  echo "=== pmc=[$pmc]" >> $O
  if [ -n "$f" ]; then
    python3 - "$f" >> $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")
    if not any(t in k for t in ("tblock_kernel", "ff_fused_kernel", "attn3_kernel", "gemm_ring_kernel<1", "true, true>(",
                                "gemm_ring_kernel<0, false, false, false, false, true, false>", "axattn", "gemm_kernel<128, 160, 2, 2, 2, 0, true, false, false, false>")): continue
    k = k.replace("(anonymous namespace)::", "")[:84]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
  else
    echo "no counter file: $(grep -i -m2 -E "error|invalid|not found|unknown" /tmp/fp.log | cut -c1-200)" >> $O
  fi
done
cat $O | cut -c1-600
