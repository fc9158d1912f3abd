#!/bin/bash
# HBM traffic of every kernel of one bench clip: rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has
# 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only; a third pass takes the raw request counters FETCH_SIZE is
# derived from (TCC_EA0_RDREQ / _32B) in case the derived counter returns no rows on this box (it did in r02j and r03e).
# Every pass keeps its log (gpurun_out/pmc_<tag>_<counter>.log) and reports how many rows it produced.
# Output: gpurun_out/<tag>_pmc_traffic.json (keyed by kernel instantiation, with the library's sha256) and .txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r04}
R=$PWD
SHA=$(python tools/lib_id.py)
CMD="python bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-roofline"
cd /tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i + 1))
  rm -rf /tmp/pmc_$i
  ( cd $R && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$i -o run -- $CMD \
      > $R/gpurun_out/pmc_${TAG}_pass$i.log 2>&1 )
  n=$(cat $(find /tmp/pmc_$i -name "*counter_collection.csv") 2>/dev/null | wc -l)
  echo "pass $i [$c]: rc=$? rows=$n" | tee -a $R/gpurun_out/pmc_${TAG}_passes.log
  tail -3 $R/gpurun_out/pmc_${TAG}_pass$i.log >> $R/gpurun_out/pmc_${TAG}_passes.log
done
python3 - "$R/gpurun_out/${TAG}_pmc_traffic" "$SHA" "$CMD" <<'PY'
import csv, glob, json, re, sys, collections
out, sha, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for i in (1, 2, 3):
    for f in glob.glob(f"/tmp/pmc_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            k = re.sub(r"\(vx_gemm_params\)$", "", k)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
res = {}
for k, d in agg.items():
    n = max(max(cnt[k].values()), 1)
    fetch = 2.0 * 1024.0 * d.get("FETCH_SIZE", 0.0) / max(cnt[k].get("FETCH_SIZE", 0), 1)
    src = "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B)"
    if fetch <= 0 and cnt[k].get("TCC_EA0_RDREQ_sum"):
        m = cnt[k]["TCC_EA0_RDREQ_sum"]
        rd, rd32 = d["TCC_EA0_RDREQ_sum"] / m, d.get("TCC_EA0_RDREQ_32B_sum", 0.0) / m
        fetch = 2.0 * ((rd - rd32) * 64.0 + rd32 * 32.0)     # the formula FETCH_SIZE is derived with, same x2 correction
        src = "TCC_EA0_RDREQ (raw), same formula and x2 correction as FETCH_SIZE"
    res[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "fetch_source": src,
              "write_bytes_per_launch": 1024.0 * d.get("WRITE_SIZE", 0.0) / max(cnt[k].get("WRITE_SIZE", 0), 1)}
json.dump({"lib_sha256": sha, "command": cmd, "kernels": res}, open(out + ".json", "w"), indent=1)
with open(out + ".txt", "w") as f:
    f.write(f"# lib_sha256={sha} command={cmd}\n")
    f.write("HBM-side traffic per launch (rocprofv3 --pmc, separate passes; read bytes doubled: gfx950 note of MI355X_MICROARCH.md)\n")
    for k, d in sorted(res.items(), key=lambda kv: -kv[1]["launches"] * (kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"])):
        f.write(f"{d['launches']:6d} x  fetch {d['fetch_bytes_per_launch']/1e6:10.2f} MB  write {d['write_bytes_per_launch']/1e6:10.2f} MB  {k[:120]}\n")
print(open(out + ".txt").read()[:3500])
PY
