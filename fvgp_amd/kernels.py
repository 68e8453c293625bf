"""Kernel selection for the native path (mirror of the plug point `kernel_function`,
fvgp/gp_prior.py:57-63,217-224).

The reference takes an arbitrary Python callable k(x1, x2, hps).  A Python callable cannot run
on the device, so the engine recognises *named* stationary kernels whose formulas are the
reference's own (fvgp/kernels.py:16-33,98-118,166-188,440-481; gp_prior.py:376-400;
gp_bo.py:115-126) and evaluates them in HIP.  Each name below is a callable object with the
reference signature k(x1, x2, hps) -> ndarray (evaluated on the GPU and copied back), so it can
also be handed to the *reference* GP as its kernel_function.

    hps layout:  *_ard : [signal variance, l_1 .. l_d]      *_iso : [signal variance, l]
"""
import numpy as np

from . import _lib


class NativeKernel:
    """A stationary kernel the HIP assembly kernel implements (fvgp_hip_kmat)."""

    def __init__(self, name, doc):
        self.name = name
        self.kernel_id = _lib.KERNEL_IDS[name]
        self.isotropic = name.endswith("_iso")
        self.__doc__ = doc

    def n_hyperparameters(self, dim):
        return 2 if self.isotropic else dim + 1

    def __call__(self, x1, x2, hps, args=None):
        from .device import default_handle
        H = default_handle()
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        K = H.empty(len(x1), len(x2) + (len(x2) & 1))
        H.kmat(self.kernel_id, H.to_device(x1), H.to_device(x2), np.asarray(hps, dtype=np.float64), K)
        H.sync()
        return K.cpu().numpy()[:, :len(x2)].copy()

    def __repr__(self):
        return f"<fvgp_amd native kernel {self.name}>"


rbf_ard = NativeKernel("rbf_ard", "hps[0] * exp(-r^2/2), r = anisotropic distance with hps[1:]")
matern32_ard = NativeKernel("matern32_ard", "the reference default kernel: hps[0] * (1+sqrt3 r) exp(-sqrt3 r)")
matern52_ard = NativeKernel("matern52_ard", "hps[0] * (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r)")
rbf_iso = NativeKernel("rbf_iso", "hps[0] * exp(-|x-x'|^2 / (2 hps[1]^2))")
matern32_iso = NativeKernel("matern32_iso", "hps[0] * (1+sqrt3 d/l) exp(-sqrt3 d/l), l = hps[1]")
matern52_iso = NativeKernel("matern52_iso", "hps[0] * (1 + sqrt5 d/l + 5 d^2/(3 l^2)) exp(-sqrt5 d/l), l = hps[1]")

NATIVE = {k.name: k for k in (rbf_ard, matern32_ard, matern52_ard, rbf_iso, matern32_iso, matern52_iso)}


def resolve(kernel_function):
    """None -> reference default (Matern-3/2 ARD, gp_prior.py:62-63); name or NativeKernel -> native;
    any other callable -> None (host slow path, the caller keeps the callable)."""
    if kernel_function is None:
        return matern32_ard
    if isinstance(kernel_function, NativeKernel):
        return kernel_function
    if isinstance(kernel_function, str):
        if kernel_function not in NATIVE:
            raise ValueError(f"unknown native kernel {kernel_function!r}; choose from {sorted(NATIVE)}")
        return NATIVE[kernel_function]
    return None


# ---------------------------------------------------------------------------------------------
# Building blocks for USER-WRITTEN host callables: code written against the reference's
# `fvgp.kernels` helper names keeps working when its kernel is handed to fvgp_amd.GP as a Python callable (the host
# slow path, N x N over PCIe).  The named kernels above never come through here -- they are assembled by fvgp_hip_kmat.
# Formulas: fvgp/kernels.py:16-33 (squared exponential), :98-118 / :166-188 (Matern 3/2, 5/2), :440-481 (distances).
# ---------------------------------------------------------------------------------------------
def get_distance_matrix(x1, x2):
    """Euclidean distances between the rows of x1 (U, D) and x2 (V, D) -> (U, V)."""
    diff = np.asarray(x1, dtype=np.float64)[:, None, :] - np.asarray(x2, dtype=np.float64)[None, :, :]
    return np.sqrt(np.einsum("uvd,uvd->uv", diff, diff))


def get_anisotropic_distance_matrix(x1, x2, hps):
    """Distances with one length scale per input dimension: sqrt(sum_k ((x1_k - x2_k) / hps_k)^2)."""
    scale = np.asarray(hps, dtype=np.float64)[:np.shape(x1)[1]]
    return get_distance_matrix(np.asarray(x1, dtype=np.float64) / scale, np.asarray(x2, dtype=np.float64) / scale)


def squared_exponential_kernel(distance, length):
    """exp(-distance^2 / (2 length^2))"""
    return np.exp(-0.5 * (distance / length) ** 2)


def exponential_kernel(distance, length):
    """exp(-distance / length)"""
    return np.exp(-distance / length)


def matern_kernel_diff1(distance, length):
    """Matern nu = 3/2: (1 + sqrt3 d/l) exp(-sqrt3 d/l)"""
    s = np.sqrt(3.0) * distance / length
    return (1.0 + s) * np.exp(-s)


def matern_kernel_diff2(distance, length):
    """Matern nu = 5/2: (1 + sqrt5 d/l + 5 d^2 / (3 l^2)) exp(-sqrt5 d/l)"""
    s = np.sqrt(5.0) * distance / length
    return (1.0 + s + s * s / 3.0) * np.exp(-s)
