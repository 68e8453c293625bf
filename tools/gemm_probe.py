"""Where does the K loop of the fp64 GEMM lose its MFMA slots?  The same kernel with parts of the loop compiled
out (option "gemm_probe": 1 no global loads / LDS writes, 2 no barrier, 4 no LDS fragment reads; sums combine).
Results are meaningless; only the times matter.  GPU box only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)
S = 8192
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
Y = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(S, S, dtype=torch.float64, device="cuda")


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


names = {0: "full kernel", 1: "no global loads / LDS writes", 2: "no barrier", 3: "no loads/writes, no barrier",
         4: "no LDS fragment reads", 5: "no loads/writes, no fragment reads", 6: "no barrier, no fragment reads",
         7: "MFMA only", 8: "full kernel without s_setprio around the MFMAs"}
names[64] = "8-byte fragment reads in the plain k order (the kernel before the permuted-k reads)"
order = (0, 64, 0, 64) if "--kperm" in sys.argv else (0, 8, 0, 8, 1, 2, 3, 4, 5, 6, 7)
if "--kperm" in sys.argv:
    H.set_option("gemm_probe", 64); H.gemm(0, 0, 0, S, S, S, 1.0, X, Y, 0.0, C); C0 = C.clone()
    H.set_option("gemm_probe", 0); C.zero_(); H.gemm(0, 0, 0, S, S, S, 1.0, X, Y, 0.0, C)
    print("max |permuted-k - plain-k| / max|C| =", float((C - C0).abs().max() / C0.abs().max()))
for v in order:
    H.set_option("gemm_probe", v)
    ms = timeit(lambda: H.gemm(0, 0, 0, S, S, S, 1.0, X, Y, 0.0, C))
    print(json.dumps({"probe": v, "what": names[v], "ms": round(ms, 3), "tflops_equiv": round(2.0 * S ** 3 / ms / 1e9, 1)}))
H.set_option("gemm_probe", 0)
