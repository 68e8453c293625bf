"""K-loop probes of the LDS-free trailing-update kernel (parts compiled out; results meaningless except probe 0)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
big = torch.randn(40960, 50048, dtype=torch.float64, device="cuda", generator=g)
for M in (24576, 40960):
    T = M // 128
    for K in (1024, 2048):
        fl = T * (T + 1) // 2 * 128 * 128 * 2.0 * K
        for wide in (0, 1):
            if wide:
                A = big[:M, :K]; C = big[:M, 4096:4096 + M]
            else:
                A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
                C = torch.zeros(M, M, dtype=torch.float64, device="cuda")
            for direct in (0, 2):
                H.set_option("gemm_direct", direct)
                ms = timeit(lambda: H.gemm(0, 0, 1, M, M, K, -1e-9, A, A, 1.0, C))
                print(json.dumps({"M": M, "K": K, "ld": 50048 if wide else K, "direct": direct, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2)}), flush=True)
            del A, C
