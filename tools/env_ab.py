"""Same-process A/B of an environment switch the library reads per call: log-likelihood evaluations alternate between the
variable set and unset.  Usage: python tools/env_ab.py FVGP_SOMETHING 8000 12000 20000"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402

warnings.simplefilter("ignore")
var = sys.argv[1]
for n in [int(a) for a in sys.argv[2:]]:
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.0, .3, .3, .3])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    res = {0: [], 1: []}
    vals = {}
    for rep in range(10 if n <= 20000 else 4):
        for on in (0, 1):
            if on: os.environ[var] = os.environ.get("AB_VALUE", "1")
            else: os.environ.pop(var, None)
            gp.log_likelihood(th * 1.01)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            vals[on] = gp.log_likelihood(th * 1.02)
            torch.cuda.synchronize(); res[on].append(1e3 * (time.perf_counter() - t0))
    os.environ.pop(var, None)
    a, b = sorted(res[0]), sorted(res[1])
    print(f"N {n}: unset min {a[0]:.3f} median {a[len(a) // 2]:.3f} ms | {var}=1 min {b[0]:.3f} median {b[len(b) // 2]:.3f} ms | values equal: {vals[0] == vals[1]}", flush=True)
