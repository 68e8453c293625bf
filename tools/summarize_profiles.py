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
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", f"{tag}_profiles")
dst = os.path.join(ROOT, "profiles")
KERNEL = "gemm_f64_kernel<0, 0, 1>"


def counter_sum(path, name, kernel=KERNEL):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
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
# the covariance assembly (north star: "rocprof HBM GB/s for K-assembly"): what kmat_kernel WROTE, from the same WRITE_SIZE pass, against
# the algorithmic 4 N (N + 1) B of the lower 128-tiles; its duration from the kernel-trace stats of the headline command
kw_kb, nk = counter_sum(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE", "kmat_kernel")
kf_kb, _ = counter_sum(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "kmat_kernel")
if nk:
    n = 50000
    npad = (n + 127) // 128 * 128
    tiles = (npad // 128) * (npad // 128 + 1) // 2
    algo = tiles * 128 * 128 * 8.0 + n * 3 * 8.0
    avg_ns = None
    for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))):
        if "kmat_kernel" in r["Name"]:
            avg_ns = float(r["AverageNs"])
    k = {"kernel": "kmat_kernel<0, 3> (N=50000 d=3 RBF, lower 128-tiles + noise on the diagonal)", "launches_counted": nk,
         "write_bytes_per_launch": kw_kb * 1024.0 / nk, "fetch_bytes_per_launch_corrected_x2": 2.0 * kf_kb * 1024.0 / nk,
         "algorithmic_bytes_per_launch": algo, "written_over_algorithmic": kw_kb * 1024.0 / nk / algo}
    if avg_ns:
        gbs = algo / (avg_ns * 1e-9) / 1e9
        k.update(avg_launch_ms_rocprof=avg_ns * 1e-6, achieved_GBps=gbs, frac_of_8000_spec=gbs / 8000.0, frac_of_6290_achievable=gbs / 6290.0)
    rec["k_assembly"] = k
json.dump(rec, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
