"""Run under `rocprofv3 --kernel-trace --output-format csv`: C2's posterior covariance (N=20k, P=1000) four times, a marker
launch (fvgp add_matrix of 1 x 1) before the last one.  With a csv argument: per-kernel summary of the last call instead.
Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python3 tools/posterior_trace.py
       python tools/posterior_trace.py DIR/trace_kernel_trace.csv"""
import csv
import os
import sys
import warnings
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import fvgp_amd
    warnings.simplefilter("ignore")
    n = 20000
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3))
    y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.0, .3, .3, .3])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    xp = np.random.default_rng(2).random((int(os.environ.get("POSTERIOR_P", "1000")), 3))
    for key, val in (kv.split("=") for kv in sys.argv[1:]):
        gp._H.set_option(key, int(val))
    import time
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.posterior_covariance(xp)
        torch.cuda.synchronize(); print("call", i, round(1e3 * (time.perf_counter() - t0), 2), "ms", flush=True)
    ab = os.environ.get("POSTERIOR_AB")          # e.g. POSTERIOR_AB=posterior_block:1024,2048 -- alternate in one process
    if ab:
        key, vals = ab.split(":")
        vals = [int(v) for v in vals.split(",")]
        res = {v: [] for v in vals}
        for rep in range(12):
            for v in vals:
                if key.startswith("FVGP_"):          # an environment switch the library reads per call: 1 = set, 0 = unset
                    if v: os.environ[key] = "1"
                    else: os.environ.pop(key, None)
                else:
                    gp._H.set_option(key, v)
                gp.posterior_covariance(xp)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gp.posterior_covariance(xp)
                torch.cuda.synchronize(); res[v].append(1e3 * (time.perf_counter() - t0))
        for v in vals:
            r = sorted(res[v])
            print(f"{key}={v}: min {r[0]:.2f} median {r[len(r) // 2]:.2f} ms")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("fvgp::", "")
    return (name[:name.index("(")] if "(" in name else name)[:48]


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # the last call starts at the last kmat launch with more than one row of workgroups
    starts = [i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"]]
    # each call assembles k (N x P) and kk (P x P): take the second to last kmat as the start
    lo = starts[-2]
    ev = rows[lo:]
    t0, t1 = ev[0]["s"], max(r["e"] for r in ev)
    print(f"last call: {len(ev)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
    agg = defaultdict(lambda: [0, 0.0])
    for r in ev:
        name = short(r["Kernel_Name"])
        key = (name, r["Queue_Id"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""))
        agg[key][0] += 1
        agg[key][1] += (r["e"] - r["s"]) / 1e3
    for key, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {key[0]:60s} q{key[1]:>2s} grid {key[2]:>7s}x{key[3]:>4s} calls {c:4d} sum {us / 1e3:8.3f} ms avg {us / c:8.1f} us")
    by_q = defaultdict(list)
    for r in ev:
        by_q[r["Queue_Id"]].append(r)
    for q, rs in by_q.items():
        busy = sum(r["e"] - r["s"] for r in rs) / 1e6
        gaps = sum(max(0, b["s"] - a["e"]) for a, b in zip(rs, rs[1:])) / 1e6
        print(f"queue {q}: {len(rs)} kernels, busy {busy:.3f} ms, gaps {gaps:.3f} ms, first start {(rs[0]['s'] - t0) / 1e6:.3f} last end {(rs[-1]['e'] - t0) / 1e6:.3f}")
    if len(sys.argv) > 2:
        for r in ev:
            print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['Queue_Id']} {r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')} {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
        summarize(sys.argv[1])
    else:
        run()
