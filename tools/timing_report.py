"""Wall times of the facade calls at the BASELINE.json config sizes (GPU box)."""
import sys, os, json, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fvgp_amd
warnings.simplefilter("ignore")

def synth(n, d, seed=20240501):
    rng = np.random.default_rng(seed); x = rng.random((n, d))
    return x, np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)

def T(f, reps=2):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 1)

out = {}
x, y = synth(20000, 3); th = np.array([1.0, .3, .3, .3])
t0 = time.perf_counter(); gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(20000, 0.01), kernel_function="rbf_ard"); out["C2 init ms"] = round((time.perf_counter() - t0) * 1e3, 1)
xp = np.random.default_rng(2).random((1000, 3))
out["C2 loglik(theta) ms"] = T(lambda: gp.log_likelihood(th * 1.01))
out["C2 posterior_mean P=1000 ms"] = T(lambda: gp.posterior_mean(xp))
out["C2 posterior_covariance P=1000 ms"] = T(lambda: gp.posterior_covariance(xp))
out["C2 gradient ms"] = T(lambda: gp.neg_log_likelihood_gradient(th * 1.01), reps=1)
del gp; torch.cuda.empty_cache()
x, y = synth(50000, 3)
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(50000, 0.01), kernel_function="matern52_ard")
out["C3 loglik(theta) ms"] = T(lambda: gp.log_likelihood(th * 1.01))
out["C3 value+gradient ms"] = T(lambda: gp.neg_log_likelihood_gradient(th * 1.01), reps=1)
out["C3 set_hyperparameters ms"] = T(lambda: gp.set_hyperparameters(th * 1.02), reps=1)
print(json.dumps(out, indent=1))
