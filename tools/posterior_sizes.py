"""Posterior covariance over the number of prediction points: the device call alone (fvgp_hip_posterior, synchronised) and the facade
call (host array out), against the flop bound n^2 p + n p^2 at the fp64 MFMA peak.
  python tools/posterior_sizes.py [N] [P,P,...] [chunk]"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fvgp_amd
from fvgp_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Ps = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1000, 2000, 4000, 6000, 10000]
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
warnings.simplefilter("ignore")
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, 0.3, 0.3, 0.3])
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard", args={"posterior_chunk": chunk})
H = gp._H
print(f"# N={n} posterior_chunk={chunk}")
for P in Ps:
    xp = np.random.default_rng(3).random((P, 3))
    bound = 1e3 * (float(n) * n * P + float(n) * P * P) / 78.6e12
    dev = None
    if P <= chunk:
        Pp = _lib.pad128(P)
        kx, S, mean = H.empty(gp._np, Pp), H.empty(Pp, Pp), H.empty(P, 1)
        xpd = H.to_device(xp)
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            H.posterior(0, gp._x_dev, th, gp._L, gp._alpha, 1, xpd, kx, mean, None, S)
            torch.cuda.synchronize(); dev = 1e3 * (time.perf_counter() - t0)
        del kx, S
    fac = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = gp.posterior_covariance(xp)
        torch.cuda.synchronize(); fac = min(fac, 1e3 * (time.perf_counter() - t0))
    print(f"P={P:6d}: device call {('%.2f' % dev) if dev else '   -'} ms, facade {fac:8.2f} ms, flop bound {bound:7.2f} ms ({bound / fac:.2f}), groups {gp._posterior_groups if P > chunk else 0}", flush=True)
