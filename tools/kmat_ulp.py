"""Worst deviation of the device covariance assembly from the oracle's numpy kernels, in units of ulp * sigma^2 (SURVEY 8c states
4 ulp sigma^2 as the bar for K entries), per kernel -- and, beside it, the deviation of BOTH from a long-double evaluation of the same
formula (kernels.py:16-33,98-118,166-188,461-481), so that the oracle's own rounding is visible.
  python tools/kmat_ulp.py [n1] [n2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fvgp_amd import _lib
from oracle import fvgp_oracle as orc

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
EPS = np.finfo(np.float64).eps
H = _lib.Handle(0)


def longdouble_kernel(name, x1, x2, theta):
    ld = np.longdouble
    x1, x2, th = x1.astype(ld), x2.astype(ld), theta.astype(ld)
    if name.endswith("ard"):
        d2 = sum(((x1[:, None, k] - x2[None, :, k]) / th[1 + k]) ** 2 for k in range(x1.shape[1]))
        r = np.sqrt(d2)
        ell = ld(1)
    else:
        r = np.sqrt(sum((x1[:, None, k] - x2[None, :, k]) ** 2 for k in range(x1.shape[1])))
        ell = th[1]
    if name.startswith("rbf"):
        return th[0] * np.exp(-(r ** 2) / (2 * ell ** 2))
    if name.startswith("matern32"):
        return th[0] * (1 + np.sqrt(ld(3)) * r / ell) * np.exp(-np.sqrt(ld(3)) * r / ell)
    return th[0] * (1 + np.sqrt(ld(5)) * r / ell + 5 * r ** 2 / (3 * ell ** 2)) * np.exp(-np.sqrt(ld(5)) * r / ell)


print(f"# K entries, {n1} x {n2} uniform points in [0,1]^d, units of ulp(1) * sigma^2 = {EPS:.3e} * theta[0]")
print("# kernel         d  theta                          device vs oracle   device vs long double   oracle vs long double")
worst = {}
for name, d in (("rbf_ard", 3), ("rbf_ard", 1), ("matern32_ard", 2), ("matern32_ard", 3), ("matern52_ard", 3), ("matern52_ard", 5),
                ("rbf_iso", 2), ("matern32_iso", 3), ("matern52_iso", 2)):
    for seed, scale in ((1, 1.0), (2, 0.1), (3, 3.0)):
        rng = np.random.default_rng(100 * seed + d)
        x1, x2 = rng.random((n1, d)), rng.random((n2, d))
        theta = np.concatenate([[0.3 + 2.0 * rng.random()], scale * (0.2 + rng.random(d if name.endswith("ard") else 1))])
        ref = orc.KERNELS[name](x1, x2, theta)
        K = H.empty(n1, n2)
        H.kmat(_lib.KERNEL_IDS[name], H.to_device(x1), H.to_device(x2), theta, K, pad=_lib.PAD_NONE)
        H.sync()
        got = K.cpu().numpy()
        ldk = longdouble_kernel(name, x1, x2, theta)
        u = EPS * theta[0]
        a = np.max(np.abs(got - ref)) / u
        b = float(np.max(np.abs(got.astype(np.longdouble) - ldk))) / u
        c = float(np.max(np.abs(ref.astype(np.longdouble) - ldk))) / u
        worst[name] = max(worst.get(name, 0.0), a)
        print(f"{name:14s} {d:2d}  {np.array2string(theta, precision=3, max_line_width=200):30s} {a:12.2f} {b:20.2f} {c:22.2f}")
print("# worst device-vs-oracle per kernel (ulp sigma^2):", {k: round(v, 2) for k, v in worst.items()})
