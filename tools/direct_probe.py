"""K-loop probes of the LDS-free trailing-update kernel (parts compiled out; results meaningless except probe 0)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
M = 16384 + 3 * 128          # 16.9 rounds of 512 workgroups
C = torch.zeros(M, M, dtype=torch.float64, device="cuda")
T = M // 128
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for K in (1024, 2048, 4096):
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    fl = T * (T + 1) // 2 * 128 * 128 * 2.0 * K
    for direct, probes in ((0, (0, 32, 5, 37)), (2, (0, 2))):
        H.set_option("gemm_direct", direct)
        for pr in probes:
            H.set_option("gemm_probe", pr)
            ms = timeit(lambda: H.gemm(0, 0, 1, M, M, K, -1.0, A, A, 1.0, C))
            print(json.dumps({"K": K, "direct": direct, "probe": pr, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2),
                              "us_per_round": round(1e3 * ms / (T * (T + 1) / 2 / 512), 1)}), flush=True)
