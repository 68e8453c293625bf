"""Two launches for counter collection: a square GEMM (K large) and one trailing SYRK (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
rng = torch.Generator(device="cuda"); rng.manual_seed(0)
for (M, K, lower) in [(8192, 8192, 0), (32768, 1024, 1)]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=rng)
    C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=rng)
    for _ in range(2):
        H.gemm(0, 0, lower, M, M, K, -1.0, A, A, 1.0, C)
    torch.cuda.synchronize()
    del A, C
