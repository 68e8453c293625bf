"""An appended GP (update_gp_data(append=True): bordering through fvgp_hip_trsm_lower, few columns against a long factor) against a
fresh GP on the concatenated data: log-likelihood, posterior mean and covariance.   python tools/append_check.py"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
import fvgp_amd
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
for n, m in ((2500, 4), (4100, 130), (2048, 1)):
    x = rng.random((n + m, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n + m)
    th = np.array([1.1, 0.3, 0.35, 0.4]); nv = np.full(n + m, 0.01)
    a = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
    a.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
    b = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    xp = rng.random((7, 3))
    la, lb = a.log_likelihood(), b.log_likelihood()
    Sa, Sb = a.posterior_covariance(xp)["S"], b.posterior_covariance(xp)["S"]
    ma, mb = a.posterior_mean(xp)["m(x)"], b.posterior_mean(xp)["m(x)"]
    print(f"n {n} + {m}: loglik rel {abs(la - lb) / abs(lb):.2e}  S {np.max(np.abs(Sa - Sb)):.2e}  mean {np.max(np.abs(ma - mb)):.2e}", flush=True)
    assert abs(la - lb) / abs(lb) < 1e-10 and np.max(np.abs(Sa - Sb)) < 1e-9 and np.max(np.abs(ma - mb)) < 1e-8
print("append ok")
