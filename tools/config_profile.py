"""One pass over the BASELINE.json configurations other than the headline, for `rocprofv3 --kernel-trace --stats`:
C2 posterior (N=20k, P=1000), C3 value + gradient (N=50k Matern-5/2), C5 (fvGP 4 x 10k).  Pick with argv[1]."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fvgp_amd
warnings.simplefilter("ignore")


def synth(n, d, seed=20240501):
    rng = np.random.default_rng(seed); x = rng.random((n, d))
    return x, np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)


which = sys.argv[1] if len(sys.argv) > 1 else "C2"
th = np.array([1.0, 0.3, 0.3, 0.3])
if which == "C2":
    x, y = synth(20000, 3)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(20000, 0.01), kernel_function="rbf_ard")
    xp = np.random.default_rng(2).random((1000, 3))
    for _ in range(3):
        gp.posterior_mean(xp); gp.posterior_covariance(xp)
elif which == "C3":
    x, y = synth(50000, 3)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(50000, 0.01), kernel_function="matern52_ard")
    for _ in range(2):
        gp.neg_log_likelihood_gradient(th * 1.01)
else:
    rng = np.random.default_rng(20240501); xm = rng.random((10000, 2)); s = xm.sum(axis=1)
    ym = np.stack([np.sin(3 * s), np.cos(3 * s), np.linalg.norm(xm, axis=1), np.sin(3 * s) * np.cos(3 * s)], axis=1) + 0.1 * rng.standard_normal((10000, 4))
    gp = fvgp_amd.fvGP(xm, ym, init_hyperparameters=np.array([1.0, 0.3, 0.3, 1.0]), noise_variances=np.full(ym.shape, 0.01))
    for _ in range(2):
        gp.log_likelihood(np.array([1.0, 0.3, 0.3, 1.0]) * 1.01)
torch.cuda.synchronize()
