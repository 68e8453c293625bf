"""Config C2 (N=20k, RBF): facade timings of log-likelihood and posterior at P prediction points (GPU box)."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402

warnings.simplefilter("ignore")
n = 20000
rng = np.random.default_rng(20240501)
x = rng.random((n, 3))
y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, .3, .3, .3])
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")


def T(f, reps=3):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 2)


print("loglik(theta) ms", T(lambda: gp.log_likelihood(th * 1.01)))
for P in (1, 2, 4, 8, 64, 1000, 4000):
    xp = np.random.default_rng(2).random((P, 3))
    print(f"P={P}: posterior_mean ms", T(lambda: gp.posterior_mean(xp)), " posterior_covariance ms", T(lambda: gp.posterior_covariance(xp)))
