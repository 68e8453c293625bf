"""Run under `rocprofv3 --pmc ...`: one long launch each of the register-only MFMA stream (calibration), the GEMM kernel on
8192^3 (K loop only: prologue / epilogue negligible) and the trailing-update shape (M = 40960 lower, K = 2048)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
out = H.empty(2048 * 256)
S = 8192
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(S, S, dtype=torch.float64, device="cuda")
M, K = 40960, 2048
A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
C2 = torch.zeros(M, M, dtype=torch.float64, device="cuda")
for rep in range(2):
    H.mfma_peak(out, 512, 8000)
    H.gemm(0, 0, 0, S, S, S, -1.0, X, X, 1.0, C)
    H.gemm(0, 0, 1, M, M, K, -1.0, A, A, 1.0, C2)
torch.cuda.synchronize()
