"""The four operand layouts of the fp64 GEMM at one size, plus a tall-K case shaped like the dtrtri update
(M = R, N = 1024, K = R).  GPU box only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


g = torch.Generator(device="cuda"); g.manual_seed(0)
S = 8192
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
Y = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(S, S, dtype=torch.float64, device="cuda")
for akm in (0, 1):
    for bnm in (0, 1):
        for beta in (0.0, 1.0):
            ms = timeit(lambda: H.gemm(akm, bnm, 0, S, S, S, 1.0, X, Y, beta, C))
            print(json.dumps({"layout": (akm, bnm), "beta": beta, "M=N=K": S, "ms": round(ms, 3), "tflops": round(2.0 * S ** 3 / ms / 1e9, 1)}))
R = 32768
A = torch.randn(R, R, dtype=torch.float64, device="cuda", generator=g)
for bnm in (0, 1):
    B = torch.randn(R, 1024, dtype=torch.float64, device="cuda", generator=g) if bnm else torch.randn(1024, R, dtype=torch.float64, device="cuda", generator=g)
    Cn = torch.zeros(R, 1024, dtype=torch.float64, device="cuda")
    ms = timeit(lambda: H.gemm(0, bnm, 0, R, 1024, R, 1.0, A, B, 0.0, Cn))
    print(json.dumps({"case": "M=R N=1024 K=R full", "bnm": bnm, "R": R, "ms": round(ms, 3), "tflops": round(2.0 * R * R * 1024 / ms / 1e9, 1)}))
