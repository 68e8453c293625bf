"""The resident panel kernel with and without the hand-off checksums (option "chain_verify"): same bits, zero mismatches.
   python tools/chain_verify_check.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = _lib.Handle(0)
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
npad = _lib.pad128(n)
xd = H.to_device(x); ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
theta = np.array([1.0, 0.3, 0.3, 0.3])
KV.fill_(float("nan"))
ref = H.loglik(0, xd, theta, V, ym, KV, alpha)
Lref = KV[:n, :n].tril().clone()
print("plain :", ref)
H.set_option("chain_verify", 1)
H.chain_verify_counts()
for t in range(reps):
    KV.fill_(float("nan"))
    out = H.loglik(0, xd, theta, V, ym, KV, alpha)
    bad, checks = H.chain_verify_counts()
    L = KV[:n, :n].tril()
    diff = (L != Lref)
    nd = int(diff.sum().item())
    where = ""
    if nd:
        idx = diff.nonzero()
        r0, c0 = int(idx[:, 0].min()), int(idx[:, 1].min())
        where = f" first differing row {r0} (block {r0 // 128}), col {c0} (block {c0 // 128}); rows blocks {sorted(set((idx[:, 0] // 128).tolist()))[:12]} cols blocks {sorted(set((idx[:, 1] // 128).tolist()))[:12]}"
    print(f"verify {t}: {out}  mismatches {bad} of {checks} comparisons; entries of L that differ: {nd}{where}")
H.set_option("chain_verify", 0)
