"""Posterior covariance at C2 (N = 20 000, P points) under the posterior's own options, same process.
   python tools/posterior_ab.py [P]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fvgp_amd
warnings.simplefilter("ignore")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 20000
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, .3, .3, .3])
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
xp = np.random.default_rng(2).random((P, 3))
H = gp._H
for cfg in ({"posterior_halves": 1, "posterior_block": 2048}, {"posterior_halves": 0, "posterior_block": 2048},
            {"posterior_halves": 1, "posterior_block": 1024}, {"posterior_halves": 0, "posterior_block": 1024}, {"posterior_halves": 1, "posterior_block": 2048}):
    for k, v in cfg.items():
        H.set_option(k, v)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); gp.posterior_covariance(xp); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(cfg, "posterior_covariance ms:", " ".join(f"{1e3 * t:.2f}" for t in ts))
