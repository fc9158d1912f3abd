"""Summarise a rocprofv3 kernel_trace.csv: per-kernel count / total / avg, with template args kept short.
    python tools/trace_summary.py kernel_trace.csv [lib_sha256 [command...]]
The first output line carries the identity of the kernel library the trace was taken with (`# lib_sha256=...`): bench.py
quotes a committed summary next to a live number only when it matches the library it has loaded."""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0.0])
spans = []
with open(sys.argv[1]) as f:
    r = csv.DictReader(f)
    for row in r:
        name = row.get("Kernel_Name") or row.get("kernel_name")
        s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = name[:110]
        spans.append((s, e, name))
        rows[name][0] += 1
        rows[name][1] += (e - s) * 1e-3
if len(sys.argv) > 2:
    print(f"# lib_sha256={sys.argv[2]} command={' '.join(sys.argv[3:])}")
tot = sum(v[1] for v in rows.values())
print(f"total kernel time {tot/1e3:.1f} ms over {sum(v[0] for v in rows.values())} launches")
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{v[1]/1e3:10.2f} ms {100*v[1]/tot:5.1f}% n={v[0]:6d} avg={v[1]/v[0]:9.1f} us  {k}")

# idle time between consecutive kernels (one stream, in order): the dispatch-to-dispatch latency of dependent launches.
# Gaps above 50 us are host stalls (model build, synchronisation points), not launch latency: listed apart.
spans.sort()
gaps = [spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1)]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]
small = sorted(g * 1e-3 for g in gaps if 0 <= g < 50_000)
big = [g * 1e-3 for g in gaps if g >= 50_000]
if small:
    print(f"# inter-kernel gaps < 50 us: n={len(small)} total {sum(small) / 1e3:.1f} ms ({100 * sum(small) / 1e3 / (tot / 1e3):.1f} % of kernel time), "
          f"median {small[len(small) // 2]:.2f} us, p90 {small[int(0.9 * len(small))]:.2f} us; gaps >= 50 us: n={len(big)} total {sum(big) / 1e3:.1f} ms; "
          f"overlapping launches: {sum(1 for g in gaps if g < 0)}")
    # where the idle time sits: gaps below 50 us by the kernel that FOLLOWS the gap (its dispatch waited) and by (before -> after)
    after, pair = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    for i, g in enumerate(gaps):
        if 0 <= g < 50_000:
            for d, k in ((after, short(spans[i + 1][2])), (pair, short(spans[i][2]) + " -> " + short(spans[i + 1][2]))):
                d[k][0] += 1
                d[k][1] += g * 1e-3
    print("# idle time by the kernel after the gap (top 12):")
    for k, v in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"#   {v[1] / 1e3:8.2f} ms n={v[0]:6d} avg={v[1] / v[0]:6.2f} us  {k}")
    print("# idle time by (kernel before -> kernel after) (top 12):")
    for k, v in sorted(pair.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"#   {v[1] / 1e3:8.2f} ms n={v[0]:6d} avg={v[1] / v[0]:6.2f} us  {k}")
