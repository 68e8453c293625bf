"""Soak of the resident panel kernel's in-launch hand-offs (chain.hip): the same evaluation repeated under look-ahead -- the panel
kernel beside a full trailing update, uneven load, consumers with warm L1 -- must give the same bits every time (a stale or torn
hand-off would change some entry of alpha), and LAPACK's answer.   python tools/chain_soak.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H = _lib.Handle(0)
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
npad = _lib.pad128(n)
xd = H.to_device(x); ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
theta = np.array([1.0, 0.3, 0.3, 0.3])
ref = None
bad = 0
for t in range(reps):
    KV.fill_(float("nan"))                                   # nothing of the previous run may be read
    out = H.loglik(0, xd, theta, V, ym, KV, alpha)
    a = alpha[:n, 0].clone()
    L = KV[:n, :n].tril().clone()
    if ref is None:
        ref = (out, a, L)
    else:
        same = out == ref[0] and torch.equal(a, ref[1]) and torch.equal(L, ref[2])
        bad += 0 if same else 1
print(f"N {n}: {reps} evaluations, {bad} differ from the first in some bit; loglik {ref[0][0]!r}")
sys.exit(1 if bad else 0)
