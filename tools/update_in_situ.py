"""The first trailing update of an N-point factorisation replayed on its own: operands in place (the factored panel inside the
matrix, C = the trailing matrix) vs the same shapes on compact random buffers -- is the in-situ rate a property of the data /
layout or of the schedule around it?   python tools/update_in_situ.py [N] [K]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = _lib.Handle(0)
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
npad = _lib.pad128(n)
xd = H.to_device(x); ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
H.loglik(0, xd, np.array([1.0, 0.3, 0.3, 0.3]), V, ym, KV, alpha)
torch.cuda.synchronize()


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


M = npad - K
T = M // 128; tiles = T * (T + 1) // 2
A = KV[K:, :K]; C = KV[K:, K:]
for name, fn in (("in situ, role 0", lambda: H.gemm(0, 0, 1, M, M, K, -1.0, A, A, 1.0, C)),
                 ("in situ, role 1", lambda: H.syrk_rowshard(M, M, K, A, A, C, 1, 0, 1, 0, 0))):
    ms = timeit(fn)
    print(json.dumps({"case": name, "N": n, "M": M, "K": K, "ms": round(ms, 3), "us_per_round512": round(1e3 * ms / (tiles / 512.0), 1)}), flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(0)
A2 = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
ms = timeit(lambda: H.gemm(0, 0, 1, M, M, K, -1.0, A2, A2, 1.0, C))
print(json.dumps({"case": "random compact A, C in situ", "M": M, "K": K, "ms": round(ms, 3), "us_per_round512": round(1e3 * ms / (tiles / 512.0), 1)}), flush=True)
A3 = A.contiguous()
ms = timeit(lambda: H.gemm(0, 0, 1, M, M, K, -1.0, A3, A3, 1.0, C))
print(json.dumps({"case": "real panel, compact copy, C in situ", "M": M, "K": K, "ms": round(ms, 3), "us_per_round512": round(1e3 * ms / (tiles / 512.0), 1)}), flush=True)
# sustained: the same launch ten times back to back (no host synchronisation in between), per-launch times from events
evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
evs[0].record()
for i in range(10):
    H.syrk_rowshard(M, M, K, A, A, C, 1, 0, 1, 0, 0)
    evs[i + 1].record()
torch.cuda.synchronize()
print(json.dumps({"case": "ten launches back to back, us per round each", "us_per_round512": [round(1e3 * evs[i].elapsed_time(evs[i + 1]) / (tiles / 512.0), 1) for i in range(10)]}), flush=True)
