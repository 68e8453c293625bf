"""One-off fuzz of the fused evaluation over problem sizes (GPU box): log-likelihood, alpha and the gradient route's inputs
against numpy / scipy on the same inputs, sizes drawn around every schedule switch (128-multiples +-1; one / two / three wide panels:
4096, 8192; sizes whose appended rows need a block row of their own; a last panel of one or two blocks), once under the default
schedule and once under look-ahead.   python tools/fuzz_sizes.py [count]"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fvgp_amd import _lib  # noqa: E402

H = _lib.Handle(0)
rng = np.random.default_rng(7)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
special = [1, 2, 127, 128, 129, 255, 257, 3968, 4095, 4096, 4097, 4223, 4224, 4225, 4352, 4607, 4608, 4609, 4700, 6143, 6144, 6145, 8064, 8191, 8192, 8193, 8320, 9000]
sizes = special[:max(0, count - 6)] + [int(v) for v in rng.integers(3, 9000, size=6)]
look = len(sys.argv) > 2 and sys.argv[2] == "lookahead"
if look:
    H.set_option("lookahead_min", 4608); H.set_option("chain_wide", 0)
worst = 0.0
for n in sizes:
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    theta = np.array([1.0, 0.3, 0.35, 0.4]); nv = np.full(n, 0.01)
    d2 = ((x[:, None, :] - x[None, :, :]) / theta[1:]) ** 2
    K = theta[0] * np.exp(-0.5 * d2.sum(-1)) + np.diag(nv)
    ym = y - y.mean()
    c = sla.cho_factor(K, lower=True)
    a = sla.cho_solve(c, ym)
    ref = -0.5 * ym @ a - np.sum(np.log(np.diag(c[0]))) - 0.5 * n * np.log(2 * np.pi)
    npad = _lib.pad128(n)
    ymd = H.zeros(npad, 1); ymd[:n, 0] = H.to_device(ym)
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    ll, logdet, quad, info = H.loglik(0, H.to_device(x), theta, H.to_device(nv), ymd, KV, alpha)
    H.sync()
    ea = float(np.max(np.abs(alpha[:n, 0].cpu().numpy() - a)) / max(np.max(np.abs(a)), 1e-300))
    el = abs(ll - ref) / abs(ref)
    worst = max(worst, ea, el)
    print(f"n {n:5d}: info {info}  rel err loglik {el:.2e}  alpha {ea:.2e}", flush=True)
    assert info == 0 and el < 1e-10 and ea < 1e-8, (n, el, ea)
print("schedule:", "look-ahead (lookahead_min 4608, chain_wide 0)" if look else "default (wide panels)", "sizes", len(sizes), "worst relative error", f"{worst:.2e}")
