"""A/B of leaf-kernel variants: wall time of a chain of 64 leaves (potrf of an 8192 matrix is chain-bound; here simply the
leaf launches of potrf on 128-row problems, run back to back) -- read the kernel's average from rocprofv3 --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
H = _lib.Handle(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 128)); M = B @ B.T + 128 * np.eye(128)
src = H.to_device(np.tril(M)); A = src.clone()
for variant in (0, 0, 0):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(40):
        A.copy_(src)
        e0.record(); H.potrf(A, 128); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"variant {variant}: potrf(128) = pad + leaf + info read-back: median {np.median(ts):.1f} us, min {min(ts):.1f} us")
