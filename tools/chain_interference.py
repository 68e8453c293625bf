"""What slows the panel chain's latency-bound kernels under look-ahead?  A chain of 128-row leaves (potrf of a 128 x 128 block:
pad + leaf + read-back) and of 2048-row panel steps is timed alone, beside a register-only MFMA stream on every CU (matrix
pipes busy, no memory traffic) and beside a streaming copy (memory system busy, no matrix work), each on another stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
H = _lib.Handle(0)
side = torch.cuda.Stream(priority=0)
Hs = _lib.Handle(0, stream=side.cuda_stream)
rng = np.random.default_rng(0)
n = 2048
B = rng.standard_normal((n, n)); M = B @ B.T + n * np.eye(n)
src = H.to_device(np.tril(M)); A = src.clone()
out = Hs.empty(4096 * 256)
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")      # 2 GiB
big2 = torch.empty_like(big)


def chain_ms(reps=5):
    ts = []
    for _ in range(reps):
        A.copy_(src); torch.cuda.current_stream().synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); H.potrf(A, n); e1.record(); torch.cuda.current_stream().synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


print("potrf(2048) alone (16 chain steps): %.3f ms" % chain_ms())
for name, load in (("register-only MFMA stream, 2 workgroups per CU", lambda: Hs.mfma_peak(out, 2048, 60000)),
                   ("streaming copy of 2 GiB (x8)", lambda: [big2.copy_(big) for _ in range(40)])):
    with torch.cuda.stream(side):
        load()
    t = chain_ms(reps=3)
    torch.cuda.synchronize()
    print("potrf(2048) beside %s: %.3f ms" % (name, t))
