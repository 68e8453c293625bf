"""Is the K loop limited by operand fetch latency?  Same GEMM with lda = ldb = 0 (every row aliases row 0:
all operand loads hit L1/L2) vs the real strides."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
L = _lib.lib()
def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
rng = torch.Generator(device="cuda"); rng.manual_seed(0)
for (M, K, lower) in [(8192, 8192, 0), (32768, 1024, 1)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=rng)
    C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=rng)
    T = M // 128
    fl = (T * (T + 1) / 2 if lower else T * T) * 128 * 128 * 2.0 * K
    for v in (0, 1):
        H.set_option("gemm_variant", v)
        for ld in (K, 0):
            for beta in (1.0, 0.0):
                f = lambda: L.fvgp_hip_gemm(H._h, 0, 0, lower, M, M, K, -1.0, ctypes.c_void_p(A.data_ptr()), ld,
                                            ctypes.c_void_p(A.data_ptr()), ld, beta, ctypes.c_void_p(C.data_ptr()), M)
                ms = timeit(f)
                print(json.dumps({"M": M, "K": K, "lower": lower, "variant": v, "ld": ld, "beta": beta, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2)}))
    del A, C
