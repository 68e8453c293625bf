"""Kernel-trace summary of the LAST `ms` milliseconds of a rocprofv3 --kernel-trace csv: per (kernel, queue) totals, the time
covered by at least one kernel, and the idle remainder -- is a schedule bound by the device or by the host that enqueues it?
  python tools/trace_busy.py trace_kernel_trace.csv 100"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:name.index("(")][:56] if "(" in name else name[:56]


rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t1 = max(r["e"] for r in rows)
t0 = t1 - int(float(sys.argv[2]) * 1e6)
ev = [r for r in rows if r["s"] >= t0]
agg = defaultdict(lambda: [0, 0.0])
for r in ev:
    a = agg[(short(r["Kernel_Name"]), r["Queue_Id"])]
    a[0] += 1
    a[1] += (r["e"] - r["s"]) / 1e6
print(f"{len(ev)} kernels in the last {(t1 - t0) / 1e6:.1f} ms")
for (k, q), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {k:56s} q{q:>2s} calls {c:6d} sum {ms:9.2f} ms avg {1e3 * ms / c:9.1f} us")
cover, end = 0, t0
for r in ev:
    if r["e"] > end:
        cover += r["e"] - max(r["s"], end)
        end = r["e"]
print(f"covered by a kernel: {cover / 1e6:.2f} ms, idle: {(t1 - t0 - cover) / 1e6:.2f} ms")
gaps = sorted(((b["s"] - a["e"]) / 1e3, short(a["Kernel_Name"]), short(b["Kernel_Name"])) for a, b in zip(ev, ev[1:]) if b["s"] > a["e"])
print("largest gaps (us, after, before):", [(round(g, 1), a[:24], b[:24]) for g, a, b in gaps[-6:]])
