"""The 256 x 128 macro-tile probe (gemm_probe 9000) against the shipped 128 x 128 kernel on full (M,K) x (N,K) products:
bitwise comparison (same k order), then interleaved timings.   python tools/macro_tile_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fvgp_amd import _lib  # noqa: E402

H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


S = 2048
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
Y = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
res = {}
for v in (0, 9000):
    C = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    H.set_option("gemm_probe", v)
    H.gemm(0, 0, 0, S, S, S, -0.5, X, Y, 1.25, C)
    torch.cuda.synchronize()
    res[v] = C.clone()
ref = -0.5 * X @ Y.T + 1.25 * torch.randn(S, S, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
print("macro tile vs shipped kernel bitwise equal:", torch.equal(res[0], res[9000]), " max |err| vs torch:", float((res[9000] - ref).abs().max()))
for (M, N, K) in [(8192, 8192, 8192), (40960, 8192, 2048), (16384, 16384, 1024)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    B = torch.randn(N, K, dtype=torch.float64, device="cuda", generator=g)
    C = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    out = []
    for rep in range(2):
        for v in (0, 9000):
            H.set_option("gemm_probe", v)
            ms = timeit(lambda: H.gemm(0, 0, 0, M, N, K, -1.0, A, B, 1.0, C))
            out.append((v, 2.0 * M * N * K / ms / 1e9))
    print(f"M {M} N {N} K {K}: " + "  ".join(f"probe {v}: {t:.1f} TFLOP/s" for v, t in out))
    del A, B, C
H.set_option("gemm_probe", 0)
