"""Config C3 (N=50k, Matern-5/2): wall time of value + gradient through the facade (GPU box)."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fvgp_amd  # noqa: E402

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rng = np.random.default_rng(20240501)
x = rng.random((n, 3))
y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, .3, .3, .3])
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="matern52_ard")
for i in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = gp.neg_log_likelihood_gradient(th * 1.01)
    torch.cuda.synchronize()
    print("value+gradient ms", round((time.perf_counter() - t0) * 1e3, 1), g)
