#!/bin/bash
# effective clock and matrix-pipe occupancy of the GEMM kernels on a few shapes (GRBM_GUI_ACTIVE / duration;
# SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles): are the long-K launches power / clock bound like the d = 40 attention?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r02n}
L=$PWD/v-express_amd/libvexpress_hip.so
B=$PWD/tools/gemm_bench
O=$PWD/gpurun_out/${T}_gemm_clock.txt
: > $O
cd /tmp
for shape in "L0 conv3x3 320>320 prepad" "L1 conv3x3 640>640 prepad" "L0 lin 320>320 +res" "L0 geglu" "L2 lin 1280>1280 +res" "L0 ffout"; do
  for pmc in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    rm -rf /tmp/pmc_out
    timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_out -o run -- $B $L 5 "$shape" > /tmp/gb.log 2>&1
    f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
    echo "=== shape=[$shape] pmc=[$pmc]" >> $O
    grep "^L" /tmp/gb.log | cut -c1-100 >> $O
    if [ -n "$f" ]; then
      python3 - "$f" >> $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
dur = collections.defaultdict(float); nd = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    if "Start_Timestamp" in r and "End_Timestamp" in r:
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); nd[k] += 1
for k, d in agg.items():
    out = {c: round(v / cnt[(k, c)]) for c, v in d.items()}
    if nd[k]: out["avg_ns"] = round(dur[k] / nd[k])
    print(k, out)
PY
    fi
  done
done
cat $O
