import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, fvgp_amd
warnings.simplefilter("ignore")
for n in (1000, 1024, 2000, 2048, 4000, 4096, 8000, 8192, 20000, 20480):
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.0, .3, .3, .3])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); gp.log_likelihood(th * (1.01 + 0.001 * i)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"N {n}: log_likelihood {1e3 * min(ts):.3f} ms", flush=True)
