"""Facade overhead: GP.log_likelihood(theta) against the bare fvgp_hip_loglik call at small N (GPU box)."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402
from fvgp_amd import _lib  # noqa: E402

warnings.simplefilter("ignore")
for n in (500, 2000, 8000):
    rng = np.random.default_rng(1)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.0, .3, .3, .3]); nv = np.full(n, 0.01)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    H = gp._H
    npad = _lib.pad128(n)
    KV = H.empty(npad, npad); al = H.empty(npad, 1)
    xd, vd, ym = H.to_device(x), H.to_device(nv), H.to_device((y - y.mean()).reshape(n, 1))

    def T(f, reps=200):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

    print(f"n={n}: facade log_likelihood(theta) {T(lambda: gp.log_likelihood(th * 1.01)):.3f} ms, bare ABI call {T(lambda: H.loglik(0, xd, th * 1.01, vd, ym, KV, al)):.3f} ms")
