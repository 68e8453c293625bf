"""A/B of K-loop variants of the fp64 GEMM through the probe switch (GPU box only): interleaved timings on the shapes the
factorisation runs, and a bitwise comparison of the results.   python tools/gemm_ab.py 0 128"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)
variants = [int(v) for v in sys.argv[1:]] or [0, 128]
g = torch.Generator(device="cuda"); g.manual_seed(0)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


# bitwise: the variants contract k in the same order
S = 2048
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
Y = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
ref = None
for v in variants:
    C = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    H.set_option("gemm_probe", v)
    H.gemm(0, 0, 0, S, S, S, -1.0, X, Y, 1.0, C)
    torch.cuda.synchronize()
    if ref is None:
        ref = C.clone()
        print("reference check vs torch:", float((C - (torch.randn(S, S, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) - X @ Y.T)).abs().max()))
    else:
        print(f"variant {v} vs {variants[0]}: max abs diff {float((C - ref).abs().max())}")
del X, Y, C, ref

for (M, N, K, lower) in [(8192, 8192, 8192, 0), (40960, 40960, 2048, 1), (20480, 20480, 1024, 1), (8192, 8192, 1024, 1)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    C = torch.randn(M, N, dtype=torch.float64, device="cuda", generator=g)
    T = M // 128
    fl = (T * (T + 1) / 2 if lower else T * T) * 128 * 128 * 2.0 * K
    for rep in range(2):
        for v in variants:
            H.set_option("gemm_probe", v)
            ms = timeit(lambda: H.gemm(0, 0, lower, M, N, K, -1.0, A, A, 1.0, C))
            print(json.dumps({"M": M, "K": K, "lower": lower, "variant": v, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2)}))
    del A, C
H.set_option("gemm_probe", 0)
