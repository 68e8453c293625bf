"""Runs a few log-likelihood evaluations at one size (for `rocprofv3 --kernel-trace --output-format csv`), or, given a
trace csv, prints the last evaluation's kernel list: per-kernel totals, idle time of the device, and the raw sequence.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o trace -- python3 tools/eval_trace.py run 8192
  python tools/eval_trace.py show gpurun_out/tr/trace_kernel_trace.csv [--seq] [--back K: start at the K-th last assembly kernel]
`runpost N P` runs posterior_covariance at P points instead (two assembly kernels per call: --back 2)."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, reps=6):
    import time
    import numpy as np
    import torch
    from fvgp_amd import _lib
    H = _lib.Handle(0)
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    xd = H.to_device(x); npad = _lib.pad128(n)
    ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
    V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    ts = []
    for t in range(reps):
        theta = np.array([1.0, 0.3, 0.3, 0.3]) * (1 + 0.02 * t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = H.loglik(0, xd, theta, V, ym, KV, alpha)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("N", n, "loglik", out[0], "ms per evaluation:", " ".join(f"{1e3 * t:.2f}" for t in ts))


def runpost(n, P, reps=4):
    import time, warnings
    import numpy as np
    import torch
    import fvgp_amd
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=np.array([1.0, .3, .3, .3]), noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    xp = rng.random((P, 3))
    ts = []
    for t in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.posterior_covariance(xp)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("N", n, "P", P, "posterior_covariance ms:", " ".join(f"{1e3 * t:.2f}" for t in ts))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if "<" in name:
        return name[:name.index("(")] if "(" in name and name.index("(") < 60 else name[:60]
    return name.split("(")[0][:50]


def show(path, seq, back=1):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"], r["k"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    starts = [i for i, r in enumerate(rows) if "kmat_kernel" in r["k"]]
    ev = rows[starts[-back]:]
    t0, t1 = ev[0]["s"], max(r["e"] for r in ev)
    print(f"last evaluation: {len(ev)} kernels, span {(t1 - t0) / 1e3:.1f} us")
    agg = defaultdict(lambda: [0, 0.0])
    for r in ev:
        a = agg[r["k"]]; a[0] += 1; a[1] += (r["e"] - r["s"]) / 1e3
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:62s} calls {c:5d}  sum {us:9.1f} us  avg {us / c:8.1f} us")
    # idle: time in [t0, t1] covered by no kernel
    cover, end = 0, t0
    for r in ev:
        if r["e"] > end:
            cover += r["e"] - max(r["s"], end); end = r["e"]
    print(f"device idle inside the evaluation: {(t1 - t0 - cover) / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f}")
    if seq:
        for r in ev:
            print(f"  {(r['s'] - t0) / 1e3:9.1f} +{(r['e'] - r['s']) / 1e3:7.1f} us  q{r['Queue_Id']:>2s} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):5d}x{r['Grid_Size_Y']:>3s}  {r['k']}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    elif sys.argv[1] == "runpost":
        runpost(int(sys.argv[2]), int(sys.argv[3]))
    else:
        back = int(sys.argv[sys.argv.index("--back") + 1]) if "--back" in sys.argv else 1
        show(sys.argv[2], "--seq" in sys.argv, back)
