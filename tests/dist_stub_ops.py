"""CPU stand-in for fvgp_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY.

Same method surface on host memory, so the partition and collective logic of the row-sharded path can run
under gloo on a machine without a GPU:
  * the sharded evaluation itself (fvgp_hip_loglik_dist) and the collective entries are the CPU twin of the C ABI
    (oracle/_cpu/libfvgp_cpu.so: the SAME driver, fvgp_amd/csrc/dist_driver.h, over host loops), bound with the
    package's own ctypes declarations;
  * the operations ShardedGP sequences itself after the factorisation (products, triangular solves, the trace pass) are
    plain torch / numpy arithmetic with the oracle's kernels.
Never imported by the package."""
import contextlib
import ctypes
import os
import subprocess

import numpy as np
import torch

from fvgp_amd import _lib
from oracle import fvgp_oracle as orc

NAMES = {0: "rbf_ard", 1: "matern32_ard", 2: "matern52_ard", 3: "rbf_iso", 4: "matern32_iso", 5: "matern52_iso"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CPU = None


def cpu_abi():
    """the CPU twin of the ABI, loaded once (built by __graft_entry__.build(); built here if missing)"""
    global _CPU
    if _CPU is None:
        path = os.path.join(ROOT, "oracle", "_cpu", "libfvgp_cpu.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
        _CPU = _lib.bind_dist(ctypes.CDLL(path))
        _CPU.fvgp_hip_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]
        _CPU.fvgp_hip_last_error_string.restype = ctypes.c_char_p
    return _CPU


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} (CPU twin) failed with status {rc}: {cpu_abi().fvgp_hip_last_error_string().decode()}")


class StubOps:
    torch = torch
    native_collectives = False

    def __init__(self):
        self._h = ctypes.c_void_p()
        _chk(cpu_abi().fvgp_hip_create(ctypes.byref(self._h), 0, None), "fvgp_hip_create")

    def stream(self):
        return contextlib.nullcontext()

    def wrap(self, ptr, count):
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_double * int(count)).from_address(int(ptr))))

    def host_sync(self):
        pass

    # -- the ABI's row-sharded entries on the CPU twin ----------------------------------------------------------------
    def dist_workspace(self, desc):
        out = (ctypes.c_int64 * 6)()
        _chk(cpu_abi().fvgp_hip_dist_workspace(ctypes.byref(desc), out), "fvgp_hip_dist_workspace")
        return list(out)

    def loglik_dist(self, desc, theta):
        t = np.ascontiguousarray(theta, dtype=np.float64)
        out = (ctypes.c_double * 3)()
        info = ctypes.c_int(0)
        _chk(cpu_abi().fvgp_hip_loglik_dist(self._h, ctypes.byref(desc), t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(t.size),
                                            out, ctypes.byref(info)), "fvgp_hip_loglik_dist")
        return out[0], out[1], out[2], info.value

    def comm_init_callbacks(self, coll, rank, nranks):
        self._coll = coll
        _chk(cpu_abi().fvgp_hip_comm_init_callbacks(self._h, ctypes.byref(coll), int(rank), int(nranks)), "fvgp_hip_comm_init_callbacks")

    def all_reduce(self, t):
        assert t.is_contiguous()
        _chk(cpu_abi().fvgp_hip_all_reduce(self._h, ctypes.c_void_p(t.data_ptr()), t.numel()), "fvgp_hip_all_reduce")

    def all_gather(self, send, recv):
        _chk(cpu_abi().fvgp_hip_all_gather(self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel()),
             "fvgp_hip_all_gather")

    def comm_profile(self):
        return {}

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or torch.float64)

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def kmat(self, kernel_id, x1, x2, theta, out, vdiag=None, pad=2):
        k = orc.KERNELS[NAMES[kernel_id]](x1.numpy(), x2.numpy(), np.asarray(theta))
        r = (len(x1) + 127) // 128 * 128
        c = (len(x2) + 127) // 128 * 128
        out[:r, :c] = 0.0
        out[:k.shape[0], :k.shape[1]] = torch.as_tensor(k)

    def gemm(self, a_kmajor, b_nmajor, lower, M, N, K, alpha, A, B, beta, C):
        """C = alpha opA opB + beta C on 128-tiles (lower: only tiles with row tile >= column tile), as fvgp_hip_gemm"""
        assert M % 128 == 0 and N % 128 == 0 and K % 16 == 0
        a = A[:K, :M].T if a_kmajor else A[:M, :K]
        b = B[:K, :N] if b_nmajor else B[:N, :K].T
        full = alpha * (a @ b)
        for ti in range(M // 128):
            for tj in range(N // 128):
                if lower and tj > ti:
                    continue
                blk = (slice(ti * 128, (ti + 1) * 128), slice(tj * 128, (tj + 1) * 128))
                C[blk] = full[blk] + (beta * C[blk] if beta != 0.0 else 0.0)

    def trsm_lower(self, L, n, B, nrhs):
        B[:n, :nrhs] = torch.linalg.solve_triangular(torch.tril(L[:n, :n]), B[:n, :nrhs], upper=False)

    def trsm_lower_t(self, L, n, B, nrhs):
        B[:n, :nrhs] = torch.linalg.solve_triangular(torch.tril(L[:n, :n]).T, B[:n, :nrhs], upper=True)

    def grad_trace_cols(self, kernel_id, x, theta, W, col0, ncols, b, partial):
        """1/2 sum over rows j >= columns k in [col0, col0 + ncols) of m_jk (W_jk - b_j b_k) dK_jk/dtheta_i, m = 1 on the
        diagonal and 2 below it (the lower triangle stands for the symmetric matrix), as fvgp_hip_grad_trace_cols"""
        n = len(x)
        xs = x.numpy()
        dK = orc.KERNEL_GRADS[NAMES[kernel_id]](xs, xs[col0:col0 + ncols], np.asarray(theta))      # (H, n, ncols)
        Ws = W[:n, :ncols].numpy().copy()
        if b is not None:
            bb = b[:n].numpy()
            Ws -= np.outer(bb, bb[col0:col0 + ncols])
        jj, kk = np.arange(n)[:, None], (col0 + np.arange(ncols))[None, :]
        m = np.where(jj > kk, 2.0, np.where(jj == kk, 1.0, 0.0))
        return np.array([0.5 * np.sum(m * Ws * dK[i]) for i in range(len(theta))])

    def add_matrix(self, A, B, alpha=1.0):
        A[:B.shape[0], :B.shape[1]] += alpha * B

    def colsumsq(self, V, out):
        out.copy_((V * V).sum(dim=0))

    def sync(self):
        pass
