import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, fvgp_amd
warnings.simplefilter("ignore")
for n in (2000, 20000):
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.0, .3, .3, .3])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    def T(f, reps=5):
        f(); torch.cuda.synchronize(); best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return round(best * 1e3, 2)
    for P in (1, 2, 4):
        xp = np.random.default_rng(2).random((P, 3))
        print(f"n={n} P={P}: mean {T(lambda: gp.posterior_mean(xp))} cov {T(lambda: gp.posterior_covariance(xp))} var-only {T(lambda: gp.posterior_covariance(xp, variance_only=True))}", flush=True)
    gp.log_likelihood(th * 1.01); xp = np.random.default_rng(2).random((1, 3))
    torch.cuda.synchronize(); t0 = time.perf_counter(); gp.posterior_covariance(xp); torch.cuda.synchronize()
    print(f"n={n} first call after a new factor: {1e3 * (time.perf_counter() - t0):.2f} ms")
