"""Run under `rocprofv3 --pmc ... --kernel-trace`: the register-only MFMA stream, the trailing-update kernel (LDS and
LDS-free) and their MFMA-only probes, one long launch each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
M, K = 16384 + 3 * 128, 4096
A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(M, M, dtype=torch.float64, device="cuda")
out = H.empty(2048 * 256)
for rep in range(2):
    H.mfma_peak(out, 1024, 10000)
    H.mfma_peak(out, 256, 20000)
    for direct, pr in ((0, 0), (0, 5), (2, 0), (2, 2)):
        H.set_option("gemm_direct", direct); H.set_option("gemm_probe", pr)
        H.gemm(0, 0, 1, M, M, K, -1.0, A, A, 0.0, C)
torch.cuda.synchronize()
