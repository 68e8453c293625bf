"""Posterior covariance right after a new factorisation (builds the inverted diagonal blocks) against the calls that follow.
  python tools/first_call_timing.py [N] [P ...]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Ps = [int(a) for a in sys.argv[2:]] or [8, 1000]
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, .3, .3, .3])
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
for P in Ps:
    xp = np.random.default_rng(2).random((P, 3))
    first, later = [], []
    for rep in range(4):
        gp.set_hyperparameters(th * (1.0 + 0.01 * (rep + 1)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.posterior_covariance(xp)
        torch.cuda.synchronize(); first.append(1e3 * (time.perf_counter() - t0))
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gp.posterior_covariance(xp)
            torch.cuda.synchronize(); later.append(1e3 * (time.perf_counter() - t0))
    print(f"N {n} P {P}: first call after a new factor {min(first[1:]):.2f} ms, later calls {min(later):.2f} ms", flush=True)
