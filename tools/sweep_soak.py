"""Soak of the single-launch backward sweep's hand-off (tagged granules) under UNEVEN load: the sweep runs beside fp64 GEMMs and
streaming copies on other streams, many times, and every result is compared bit for bit with the per-block step kernels.
  python tools/sweep_soak.py [n] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fvgp_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H = _lib.Handle(0)
npad = _lib.pad128(n)
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(n)
A = torch.zeros(npad, npad, dtype=torch.float64, device=dev)
for r0 in range(0, npad, 4096):
    r1 = min(npad, r0 + 4096)
    A[r0:r1, :r1] = 0.02 * torch.randn(r1 - r0, r1, dtype=torch.float64, device=dev, generator=g) / np.sqrt(npad)
A.diagonal().copy_(1.0 + torch.rand(npad, dtype=torch.float64, device=dev, generator=g))
A[n:, :] = 0.0
A.diagonal()[n:] = 1.0
H.invalidate_factor()
rhs = torch.randn(npad, 1, dtype=torch.float64, device=dev, generator=g)
rhs[n:] = 0.0


def solve(mode):
    H.set_option("bwd_sweep", mode)
    B = rhs.clone()
    H.potrs(A, n, B, 1)
    H.sync()
    return B[:n, 0].clone()


ref = solve(0)
assert torch.isfinite(ref).all()
side = [torch.cuda.Stream() for _ in range(3)]
X = torch.randn(4096, 4096, dtype=torch.float64, device=dev, generator=g)
big = torch.empty(64 << 20, dtype=torch.float64, device=dev)
bad = 0
for it in range(reps):
    kind = it % 4
    if kind >= 1:          # load on other streams while the sweep runs: MFMA work, a streaming copy, or both
        with torch.cuda.stream(side[0]):
            if kind in (1, 3):
                for _ in range(3): X @ X
        with torch.cuda.stream(side[1]):
            if kind in (2, 3):
                big.copy_(big.roll(1 << 20))
    got = solve(1)
    torch.cuda.synchronize()
    if not torch.equal(got, ref):
        bad += 1
        print(f"iteration {it}: {int((got != ref).sum())} entries differ, max |diff| {float((got - ref).abs().max()):.3e}", flush=True)
print(f"n {n}: {reps} sweeps beside other streams' work, {bad} differed from the step kernels")
sys.exit(1 if bad else 0)
