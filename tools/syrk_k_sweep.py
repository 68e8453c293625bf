"""Trailing-update kernel alone: time per round of 512 workgroups as a function of K, with and without the C
read-modify-write, on a compact buffer and on a window of a 50048-wide one (the in-situ leading dimension)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)


COLD = len(sys.argv) > 1 and sys.argv[1] == "cold"      # a 1 GB write between the repetitions: operands from HBM, not from the Infinity Cache
_flush = torch.empty(1 << 27, dtype=torch.float64, device="cuda") if COLD else None


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        if COLD:
            _flush.fill_(1.0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


g = torch.Generator(device="cuda"); g.manual_seed(0)
big = torch.randn(24576 + 4096, 50048, dtype=torch.float64, device="cuda", generator=g)
for M in (24576, 16384):
    for wide in (0,):
        for K in (512, 1024, 2048):
            for beta in (1.0,):
                if wide:
                    A = big[:M, :K]; C = big[:M, 4096:4096 + M]
                else:
                    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
                    C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=g)
                ms = timeit(lambda: H.gemm(0, 0, 1, M, M, K, -1.0, A, A, beta, C))
                ms1 = timeit(lambda: H.syrk_rowshard(M, M, K, A, A, C, 1, 0, 1, 0, 0)) if beta == 1.0 else float("nan")      # the same tiles as ROLE 1: yield poll in the K loop
                T = M // 128; tiles = T * (T + 1) // 2
                fl = tiles * 128 * 128 * 2.0 * K
                print(json.dumps({"M": M, "K": K, "beta": beta, "ld": 50048 if wide else M, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2),
                                  "us_per_round512": round(1e3 * ms / (tiles / 512.0), 1),
                                  "role1_us_per_round512": round(1e3 * ms1 / (tiles / 512.0), 1)}), flush=True)
                if not wide:
                    del A, C
