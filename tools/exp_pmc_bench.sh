#!/bin/bash
# HBM traffic of every kernel of one bench clip: rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has
# 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.  Output: gpurun_out/pmc_traffic_<tag>.json/.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
R=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd $R && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o run -- \
      python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /tmp/pmc_$c.log 2>&1 )
done
python3 - "$R/gpurun_out/pmc_traffic_$TAG" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"]
            agg[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE": agg[k]["n"] += 1
res = {}
for k, d in agg.items():
    n = max(d["n"], 1)
    # rocprofv3 reports both in KiB-like units of 1 KB; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2
    res[k] = {"launches": d["n"], "fetch_kb_raw_per_launch": d["FETCH_SIZE"] / n,
              "fetch_bytes_per_launch": 2.0 * 1024.0 * d["FETCH_SIZE"] / n,
              "write_bytes_per_launch": 1024.0 * d["WRITE_SIZE"] / n}
json.dump(res, open(out + ".json", "w"), indent=1)
with open(out + ".txt", "w") as f:
    f.write("HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled: gfx950 note)\n")
    for k, d in sorted(res.items(), key=lambda kv: -kv[1]["launches"] * (kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"])):
        f.write(f"{d['launches']:6d} x  fetch {d['fetch_bytes_per_launch']/1e6:10.2f} MB  write {d['write_bytes_per_launch']/1e6:10.2f} MB  {k[:110]}\n")
print(open(out + ".txt").read()[:3000])
PY
