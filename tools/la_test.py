import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
from fvgp_amd import _lib
H = _lib.Handle(0)
for n, nb in [(1537, 256), (3000, 512), (5000, 1024)]:
    rng = np.random.default_rng(n); B = rng.standard_normal((n, n)); M = B @ B.T + n * np.eye(n)
    Lref = np.tril(sla.cho_factor(M, lower=True)[0])
    for la in (0, 1):
        H.set_option("outer_block", nb); H.set_option("lookahead", la)
        npad = _lib.pad128(n); buf = np.zeros((npad, npad)); buf[:n, :n] = np.tril(M)
        A = H.to_device(buf)
        info = H.potrf(A, n)
        L = np.tril(A.cpu().numpy()[:n, :n])
        print(n, nb, "lookahead", la, "info", info, "relerr", np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)))
