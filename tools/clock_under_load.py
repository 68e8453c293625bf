"""Effective shader clock under the register-only MFMA stream and under the two trailing-update kernels: run under
`rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` and divide the counter by 8 (XCDs) and the kernel's duration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
M, K = 24576, 4096
A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=g)
out = H.empty(2048 * 256)
for rep in range(6):
    H.mfma_peak(out, 1024, 40000)                      # ~ 35 ms of register-only fp64 MFMA
    H.set_option("gemm_direct", 0)
    H.gemm(0, 0, 1, M, M, K, -1e-6, A, A, 1.0, C)
    H.set_option("gemm_direct", 2)
    H.gemm(0, 0, 1, M, M, K, -1e-6, A, A, 1.0, C)
torch.cuda.synchronize()
