"""Turn what tools/collect_profiles.sh left under gpurun_out/<TAG>_profiles into the small, tracked files under profiles/:
the bench lines, the rocprofv3 --stats kernel summaries (headline + C2 / C3 / C5) and the trailing update's HBM-side traffic
from the two --pmc passes (FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE).
Usage: python tools/summarize_profiles.py r02"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", f"{tag}_profiles")
dst = os.path.join(ROOT, "profiles")
KERNEL = "gemm_f64_kernel<0, 0, 1>"


def counter_sum(path, name):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


for f, out in (("bench_n50k_steps5.json", f"{tag}_bench_n50k_steps5.json"),
               ("bench_n50k_steps5_under_rocprof.json", f"{tag}_bench_n50k_steps5_under_rocprof.json")):
    lines = [l for l in open(os.path.join(src, f)) if l.startswith("{")]
    open(os.path.join(dst, out), "w").write(lines[-1])
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_n50k_steps5_kernel_stats.csv"))
for c in ("C2", "C3", "C5"):
    shutil.copy(os.path.join(src, f"stats_{c}", f"{c}_kernel_stats.csv"), os.path.join(dst, f"{tag}_{c}_kernel_stats.csv"))
fetch_kb, nf = counter_sum(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write_kb, nw = counter_sum(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
evals = 2                                    # --steps 1 --warmup 1
fetch_b, write_b = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
rec = {"kernel": KERNEL,
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs (two separate passes, no tracing)",
       "launches_counted": nf, "FETCH_SIZE_KB_sum_raw": fetch_kb, "fetch_bytes_corrected_x2": fetch_b,
       "WRITE_SIZE_KB_sum": write_kb, "write_bytes": write_b,
       "note": "gfx950 FETCH_SIZE counts 64 B per 128 B request: doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE exact for streaming "
               "stores. Fabric-side counters: Infinity-Cache hits are included, so this is an upper bound on HBM bytes.",
       "bytes_per_launch": (fetch_b + write_b) / max(nf, 1), "bytes_per_evaluation": (fetch_b + write_b) / evals}
assert nf == nw, (nf, nw)
l2 = os.path.join(src, "pmc_l2", "l2_counter_collection.csv")
if os.path.exists(l2):
    hit, _ = counter_sum(l2, "TCC_HIT_sum")
    miss, _ = counter_sum(l2, "TCC_MISS_sum")
    if hit + miss > 0:
        rec["l2_hit_rate"] = hit / (hit + miss)
        rec["l2_note"] = "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) over the same launches, a third --pmc pass"
json.dump(rec, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
