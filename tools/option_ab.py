"""Same-box A/B of one library option on the log-likelihood evaluation: alternates the values round by round in ONE process.
  python tools/option_ab.py leaf_yield 0,1 8000,20000,50000 [reps]
  python tools/option_ab.py panel_chain=0/panel_chain=1,panel_chain_min=0 - 20000     (whole configurations, '/'-separated)
Prints the median and the minimum of the evaluation time per (size, value)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from fvgp_amd import _lib
    key = sys.argv[1]
    if "=" in key:
        values = key.split("/")
        key = "cfg"
    else:
        values = [int(v) for v in sys.argv[2].split(",")]
    sizes = [int(v) for v in sys.argv[3].split(",")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    H = _lib.Handle(0)
    for n in sizes:
        rng = np.random.default_rng(20240501)
        x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
        xd = H.to_device(x); npad = _lib.pad128(n)
        ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
        V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
        ts = {v: [] for v in values}
        out = {}
        for t in range(reps + 1):
            for v in values:
                if key == "cfg":
                    for kv in v.split(","):
                        H.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                else:
                    H.set_option(key, v)
                theta = np.array([1.0, 0.3, 0.3, 0.3])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out[v] = H.loglik(0, xd, theta, V, ym, KV, alpha)
                torch.cuda.synchronize()
                if t > 0: ts[v].append(time.perf_counter() - t0)
        for v in values:
            a = np.array(ts[v]) * 1e3
            print(f"N {n} {key}={v}: median {np.median(a):.2f} ms  min {a.min():.2f} ms  loglik {out[v][0]!r}", flush=True)
        del KV, alpha
        torch.cuda.empty_cache()


main()
