import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, time
from fvgp_amd import _lib
H = _lib.Handle(0)
for n in (20000, 50000):
    rng = np.random.default_rng(1); x = rng.random((n, 3)); y = np.sin(3 * x.sum(1))
    npad = _lib.pad128(n)
    xd = H.to_device(x); vd = H.to_device(np.full(n, 0.01)); ymd = H.to_device((y - y.mean()).reshape(n, 1))
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    th = np.array([1.0, .3, .3, .3])
    H.loglik(0, xd, th, vd, ymd, KV, alpha)
    for prof in (0, 1):
        H.set_option("profile", prof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        H.loglik(0, xd, th * 1.01, vd, ymd, KV, alpha)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(n, "profile", prof, f"{1e3*dt:.1f} ms", H.get_profile() if prof else "")
    H.set_option("profile", 0)
    del KV, alpha
