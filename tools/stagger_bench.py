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
cases = [(8192, 8192, 0), (32768, 1024, 1)]
configs = [(256, 0, 0), (1, 99, 1024), (1, 99, 2048), (1, 99, 3072), (1, 99, 4096), (1, 99, 6144), (1, 99, 300)]
for (M, K, lower) in cases:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=rng)
    C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=rng)
    T = M // 128
    fl = (T * (T + 1) / 2 if lower else T * T) * 128 * 128 * 2.0 * K
    for (a, mod, unit) in configs:
        H.set_option("stagger_a", a); H.set_option("stagger_mod", mod); H.set_option("stagger_unit", unit)
        ms = timeit(lambda: H.gemm(0, 0, lower, M, M, K, -1.0, A, A, 1.0, C))
        print(json.dumps({"M": M, "K": K, "lower": lower, "stag": [a, mod, unit], "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2)}))
    del A, C
