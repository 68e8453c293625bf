"""update_gp_data(append=True) of a few points to a long factor: wall time per append (bordering in place) against a new factorisation.
   python tools/append_timing.py [N] [m]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fvgp_amd
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(1)
steps = 8
x = rng.random((n + steps * m, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n + steps * m)
th = np.array([1.0, 0.3, 0.3, 0.3]); nv = np.full(n + steps * m, 0.01)
gp = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
torch.cuda.synchronize(); t0 = time.perf_counter(); gp.set_hyperparameters(th); torch.cuda.synchronize()
print(f"N {n}: a new factorisation (set_hyperparameters) {1e3 * (time.perf_counter() - t0):.2f} ms")
ts = []
k = n
for s in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gp.update_gp_data(x[k:k + m], y[k:k + m], noise_variances_new=nv[k:k + m], append=True)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0)); k += m
print(f"N {n}: append of {m} points, {steps} in a row: " + " ".join(f"{t:.2f}" for t in ts) + f" ms (best {min(ts):.2f})")
