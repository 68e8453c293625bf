"""Per-workgroup start / end times of the chain's trsm_tiles launches during one evaluation (diagnostic option "chain_stamps"):
is a launch slow under look-ahead because its workgroups START late (no free slot) or because they RUN slowly?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
H = _lib.Handle(0)
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
xd = H.to_device(x); npad = _lib.pad128(n)
ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
theta = np.array([1.0, 0.3, 0.3, 0.3])
H.loglik(0, xd, theta, V, ym, KV, alpha)
stamps = torch.zeros(8 + 4 * (1 << 20), dtype=torch.int64, device="cuda")
H.set_option("chain_stamps", stamps.data_ptr())
H.loglik(0, xd, theta * 1.01, V, ym, KV, alpha)
torch.cuda.synchronize()
H.set_option("chain_stamps", 0)
s = stamps.cpu().numpy()
cnt = int(s[0]); e = s[8:8 + 4 * cnt].reshape(cnt, 4)
print("workgroups recorded", cnt)
print(" launch   WGs  first->last start us   WG run us (median / max)   launch span us")
for q in np.unique(e[:, 0]):
    w = e[e[:, 0] == q]
    t0, t1 = w[:, 2], w[:, 3]
    run = (t1 - t0) / 100.0
    print(f"{int(q):7d} {len(w):5d} {(t0.max() - t0.min()) / 100.0:12.1f} {np.median(run):18.1f} / {run.max():6.1f} {(t1.max() - t0.min()) / 100.0:14.1f}")
