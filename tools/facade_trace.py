"""Run under `rocprofv3 --kernel-trace --output-format csv`: a few calls of one facade method at one size, to look at the last call's
kernel sequence with tools/posterior_trace.py's lister.   python tools/facade_trace.py grad|post_grad|append|fvgp_post N"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402

warnings.simplefilter("ignore")
what, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, .3, .3, .3])
if what == "fvgp_post":
    xm = rng.random((n // 4, 2)); ym = np.stack([np.sin(xm.sum(1)), np.cos(xm.sum(1)), np.linalg.norm(xm, axis=1), np.sin(xm.sum(1)) * np.cos(xm.sum(1))], axis=1)
    gp = fvgp_amd.fvGP(xm, ym + 0.1 * rng.standard_normal(ym.shape), init_hyperparameters=np.array([1.0, .3, .3, 1.0]))
else:
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="matern52_ard" if what == "grad" else "rbf_ard")
xp = np.random.default_rng(2).random((64, 3 if what != "fvgp_post" else 2))
calls = {"grad": lambda: gp.neg_log_likelihood_gradient(th * 1.01),
         "post_grad": lambda: gp.posterior_mean_grad(xp[:8]) if hasattr(gp, "posterior_mean_grad") else gp.posterior_mean(xp),
         "append": lambda: gp.update_gp_data(rng.random((4, 3)), rng.standard_normal(4), noise_variances_new=np.full(4, 0.01), append=True),
         "fvgp_post": lambda: gp.posterior_covariance(xp, x_out=np.arange(4))}
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    calls[what]()
    torch.cuda.synchronize(); print(what, n, "call", i, round(1e3 * (time.perf_counter() - t0), 3), "ms", flush=True)
