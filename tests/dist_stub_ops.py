"""torch-CPU stand-in for fvgp_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY.

Same method surface, plain torch/numpy arithmetic (kernels from the oracle), so the partition and
collective logic of ShardedGP can run under gloo on a machine without a GPU.  Never imported by
the package."""
import contextlib

import numpy as np
import torch

from oracle import fvgp_oracle as orc

NAMES = {0: "rbf_ard", 1: "matern32_ard", 2: "matern52_ard", 3: "rbf_iso", 4: "matern32_iso", 5: "matern52_iso"}


class StubOps:
    torch = torch

    def __init__(self):
        self.chain = self                      # no streams on the CPU: the chain runs in program order

    def stream(self):
        return contextlib.nullcontext()

    def fork(self):
        pass

    def join(self):
        pass

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or torch.float64)

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def kmat_rows(self, kernel_id, x_rows, x_all, theta, out):
        k = orc.KERNELS[NAMES[kernel_id]](x_rows.numpy(), x_all.numpy(), np.asarray(theta))
        r = (len(x_rows) + 127) // 128 * 128
        c = (len(x_all) + 127) // 128 * 128
        out[:r, :c] = 0.0
        out[:k.shape[0], :k.shape[1]] = torch.as_tensor(k)

    def panel_potrf_dev(self, T, w, rows, n_valid, info_dev, logdet_dev):
        D = T[:w, :w]
        M = torch.tril(D) + torch.tril(D, -1).T
        L, info = torch.linalg.cholesky_ex(M)
        info_dev[0] = int(info)
        if int(info) != 0:
            T[:rows, :w] = float("nan")
            return
        T[:w, :w] = torch.tril(L) + torch.triu(D, 1)               # strict upper left as is (unspecified)
        logdet_dev[0] = 2.0 * torch.log(torch.diagonal(L)[:n_valid]).sum()
        if rows > w:
            T[w:rows, :w] = torch.linalg.solve_triangular(torch.tril(L), T[w:rows, :w].T, upper=False).T

    def syrk_rowshard(self, M, N, K, A, B, C, scale, off, b_ranks=1, b_blocks=0, b_off=0):
        for ti in range(M // 128):
            for tj in range(N // 128):
                if tj <= ti * scale + off:
                    idx = tj + b_off
                    rb = (idx % b_ranks) * b_blocks + idx // b_ranks
                    C[ti * 128:(ti + 1) * 128, tj * 128:(tj + 1) * 128] -= A[ti * 128:(ti + 1) * 128, :K] @ B[rb * 128:(rb + 1) * 128, :K].T

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

    def grad_trace(self, kernel_id, x, theta, W, b, partial):
        """1/2 sum_jk (W_jk - b_j b_k) dK_jk/dtheta_i with W symmetric, its lower triangle read (as fvgp_hip_grad_trace)"""
        n = len(x)
        Wl = torch.tril(W[:n, :n])
        Ws = (Wl + torch.tril(Wl, -1).T).numpy()
        if b is not None:
            bb = b[:n].numpy()
            Ws = Ws - np.outer(bb, bb)
        dK = orc.KERNEL_GRADS[NAMES[kernel_id]](x.numpy(), x.numpy(), np.asarray(theta))
        return np.array([0.5 * np.sum(Ws * dK[i]) for i in range(len(theta))])

    def sync(self):
        pass
