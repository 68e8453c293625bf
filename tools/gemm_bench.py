"""Micro-benchmarks of the fp64 MFMA GEMM/SYRK kernel and the raw MFMA ceiling (GPU box only)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


out = H.empty(2048 * 256)
for blocks in (512, 1024):
    iters = 4000
    ms = timeit(lambda: H.mfma_peak(out, blocks, iters))
    fl = blocks * 4 * iters * 16 * 2048.0
    print(json.dumps({"test": "mfma_peak", "blocks": blocks, "ms": ms, "tflops": fl / ms / 1e9}))

rng = torch.Generator(device="cuda"); rng.manual_seed(0)
for (M, K, lower) in [(8192, 8192, 0), (16384, 512, 1), (16384, 1024, 1), (16384, 2048, 1), (32768, 512, 1), (32768, 1024, 1),
                      (40960, 512, 1), (40960, 1024, 1), (8192, 512, 1), (4096, 512, 1), (2048, 512, 1)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=rng)
    C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=rng)
    ms = timeit(lambda: H.gemm(0, 0, lower, M, M, K, -1.0, A, A, 1.0, C), reps=3)
    T = M // 128
    fl = (T * (T + 1) / 2 if lower else T * T) * 128 * 128 * 2.0 * K
    print(json.dumps({"test": "syrk" if lower else "gemm", "M": M, "K": K, "ms": ms, "tflops": fl / ms / 1e9}))
    del A, C
