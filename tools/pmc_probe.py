"""Run under `rocprofv3 --pmc ...`: the 8192^3 GEMM once per K-loop variant (probe numbers on the command line)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
S = 8192
X = torch.randn(S, S, dtype=torch.float64, device="cuda", generator=g)
C = torch.zeros(S, S, dtype=torch.float64, device="cuda")
for rep in range(2):
    for v in [int(a) for a in sys.argv[1:]]:
        H.set_option("gemm_probe", v)
        H.gemm(0, 0, 0, S, S, S, -1.0, X, X, 1.0, C)
torch.cuda.synchronize()
