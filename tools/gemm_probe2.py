"""K-loop timing probes of the zero-VALU loop (GPU box only): 512 full, +1 no global loads / LDS writes, +2 no barrier, +4 no LDS
reads; 1536 = full loop at one workgroup per CU.  Results of the probes are meaningless; only the times matter."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
S = 8192
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(S, S, dtype=torch.float64, device="cuda")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for v in [int(a) for a in sys.argv[1:]] or [512, 513, 514, 515, 516, 517, 518, 519, 1536, 512]:
    H.set_option("gemm_probe", v)
    ms = timeit(lambda: H.gemm(0, 0, 0, S, S, S, 1.0, X, X, 0.0, C))
    print(json.dumps({"probe": v, "ms": round(ms, 3), "tflops_equiv": round(2.0 * S ** 3 / ms / 1e9, 2)}))
H.set_option("gemm_probe", 0)
