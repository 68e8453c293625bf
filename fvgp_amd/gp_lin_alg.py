"""Dense linear algebra of the hot path on the device -- mirror of the free functions in
fvgp/gp_lin_alg.py:237-360 (calculate_Chol_factor / _solve / _logdet) plus the error type
(:27-49).  ndarray in, device-resident factor object out; these three also fit the reference's
`linalg_mode=[f_factor, f_solve, f_logdet]` plug point (fvgp/gp_kv.py:457-458,552-554,625-628).
"""
import numpy as np

from . import _lib
from .device import default_handle


class NonPositiveDefiniteError(np.linalg.LinAlgError):
    """Covariance matrix is not positive definite (fvgp/gp_lin_alg.py:27-29)."""


def _non_pd_message(n, info, diag_min=None, sym_err=None):
    """Same diagnostic content as fvgp/gp_lin_alg.py:32-49, with dpotrf's info."""
    diag = "" if diag_min is None else (f"Diagnostics: min(diag(M)) = {diag_min:.3e}, "
                                        f"max|M - M.T| = {sym_err:.3e} (should be ~0).\n")
    return (f"Cholesky factorization failed: the {n}x{n} prior covariance matrix is not positive definite.\n"
            f"Most common causes in fvGP:\n"
            f"  1. A user-defined kernel that is not positive definite for all inputs.\n"
            f"  2. Duplicate or near-duplicate rows in x_data causing a rank-deficient K.\n"
            f"  3. Noise/jitter on the diagonal is too small for the conditioning of K.\n"
            f"{diag}"
            f"Try: (a) verify the kernel is PD, (b) add jitter to the diagonal, (c) deduplicate x_data.\n"
            f"Original linear-algebra error: {info}-th leading minor of the array is not positive definite")


class CholFactor:
    """Lower Cholesky factor resident in HBM (padded to 128; strict upper unspecified)."""

    def __init__(self, handle, L, n):
        self.handle, self.L, self.n = handle, L, int(n)

    def lower(self):
        """tril(L) as ndarray (what np.tril(cho_factor(M, lower=True)[0]) gives)."""
        self.handle.sync()
        return np.tril(self.L[:self.n, :self.n].cpu().numpy())


def calculate_Chol_factor(M, compute_device="gpu", args=None):
    """L L^T = M on the device.  M: (n,n) ndarray, lower triangle read."""
    assert isinstance(M, np.ndarray), "M must be np.ndarray for Cholesky factorization"
    if compute_device != "gpu":
        raise Exception("No valid compute device found. fvgp_amd computes on the MI355X only ('gpu').")
    H = default_handle()
    n = M.shape[0]
    npad = _lib.pad128(n)
    A = H.zeros(npad, npad)
    A[:n, :n] = H.to_device(M)
    info = H.potrf(A, n)
    if info != 0:
        raise NonPositiveDefiniteError(_non_pd_message(n, info, float(np.min(np.diag(M))),
                                                       float(np.max(np.abs(M - M.T)))))
    return CholFactor(H, A, n)


def calculate_Chol_solve(factor, vec, compute_device="gpu", args=None):
    """x = M^-1 vec from the factor; 1-d vec -> (n,1) (fvgp/gp_lin_alg.py:292,327)."""
    assert isinstance(vec, np.ndarray), "vec must be np.ndarray for Cholesky solve"
    if np.ndim(vec) == 1:
        vec = vec.reshape(len(vec), 1)
    vec = vec.astype(np.float64, copy=False)
    H, n = factor.handle, factor.n
    npad = _lib.pad128(n)
    c = vec.shape[1]
    cp = c if c <= _lib.MAX_RHS_VEC else _lib.pad128(c)
    B = H.zeros(npad, cp + (cp & 1) if cp > _lib.MAX_RHS_VEC else cp)
    B[:n, :c] = H.to_device(vec)
    H.potrs(factor.L, n, B, cp)
    H.sync()
    return B[:n, :c].cpu().numpy()


def calculate_Chol_logdet(factor, compute_device="gpu", args=None):
    """2 * sum(log|L_ii|) (fvgp/gp_lin_alg.py:337-338)."""
    return float(factor.handle.logdet(factor.L, factor.n))
