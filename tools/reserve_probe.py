"""Does the panel chain gain from compute units of its own?  The handle's main stream (trailing updates) is restricted to
256 - R compute units (R/8 taken from every XCD); the look-ahead stream stays unrestricted, so the single-workgroup leaf
lands on a unit it does not share with trailing-update waves (which use the same fp64 pipes).  GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fvgp_amd import _lib


def run(n, reserve, reps=5):
    stream = None
    if reserve:
        stream = _lib.create_stream(0, cu_mask=list(range(0, 256 - reserve)))
    H = _lib.Handle(0, stream=stream)
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    xd = H.to_device(x); npad = _lib.pad128(n)
    ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
    V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    torch.cuda.synchronize()
    ts = []
    for t in range(reps):
        theta = np.array([1.0, 0.3, 0.3, 0.3]) * (1 + 0.02 * t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = H.loglik(0, xd, theta, V, ym, KV, alpha)
        H.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"N {n} reserve {reserve}: loglik {out[0]:.6f}  ms: " + " ".join(f"{1e3 * t:.2f}" for t in ts), flush=True)
    H.close()
    if stream is not None:
        _lib.destroy_stream(stream)


for n in [int(a) for a in sys.argv[1].split(",")]:
    for r in [int(a) for a in sys.argv[2].split(",")]:
        run(n, r)
