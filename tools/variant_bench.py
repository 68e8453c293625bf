import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
rng = torch.Generator(device="cuda"); rng.manual_seed(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
for (M, K, lower) in [(8192, 8192, 0), (32768, 1024, 1), (32768, 512, 1), (16384, 1024, 1), (8192, 1024, 1), (4096, 1024, 1)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=rng)
    C0 = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=rng)
    T = M // 128
    fl = (T * (T + 1) / 2 if lower else T * T) * 128 * 128 * 2.0 * K
    ref = None
    for v in variants:
        H.set_option("gemm_variant", v)
        C = C0.clone()
        H.gemm(0, 0, lower, M, M, K, -1.0, A, A, 1.0, C); torch.cuda.synchronize()
        if ref is None: ref = C
        err = float((C - ref).abs().max())
        ms = timeit(lambda: H.gemm(0, 0, lower, M, M, K, -1.0, A, A, 1.0, C))
        print(json.dumps({"M": M, "K": K, "lower": lower, "variant": v, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2), "maxdiff_vs_v0": err}))
    del A, C0, C, ref
