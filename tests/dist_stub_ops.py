"""torch-CPU stand-in for fvgp_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY.

Same method surface, plain torch/numpy arithmetic (kernels from the oracle), so the partition and
collective logic of ShardedGP can run under gloo on a machine without a GPU.  Never imported by
the package."""
import numpy as np
import torch

from oracle import fvgp_oracle as orc

NAMES = {0: "rbf_ard", 1: "matern32_ard", 2: "matern52_ard", 3: "rbf_iso", 4: "matern32_iso", 5: "matern52_iso"}


class StubOps:
    torch = torch

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def kmat_rows(self, kernel_id, x_rows, x_all, theta, out):
        k = orc.KERNELS[NAMES[kernel_id]](x_rows.numpy(), x_all.numpy(), np.asarray(theta))
        r = (len(x_rows) + 127) // 128 * 128
        c = (len(x_all) + 127) // 128 * 128
        out[:r, :c] = 0.0
        out[:k.shape[0], :k.shape[1]] = torch.as_tensor(k)

    def potrf(self, D, n):
        L, info = torch.linalg.cholesky_ex(torch.tril(D[:n, :n]) + torch.tril(D[:n, :n], -1).T)
        if int(info) != 0:
            return int(info)
        D[:n, :n] = torch.tril(L) + torch.triu(D[:n, :n], 1)       # strict upper left as is (unspecified)
        return 0

    def panel_trsm(self, D, nd, Pm, rows):
        L = torch.tril(D[:nd, :nd])
        Pm[:rows, :nd] = torch.linalg.solve_triangular(L, Pm[:rows, :nd].T, upper=False).T

    def syrk_rowshard(self, M, N, K, A, B, C, scale, off):
        for ti in range(M // 128):
            for tj in range(N // 128):
                if tj <= ti * scale + off:
                    C[ti * 128:(ti + 1) * 128, tj * 128:(tj + 1) * 128] -= A[ti * 128:(ti + 1) * 128, :K] @ B[tj * 128:(tj + 1) * 128, :K].T

    def trsm_lower(self, D, n, B, nrhs):
        B[:n, :nrhs] = torch.linalg.solve_triangular(torch.tril(D[:n, :n]), B[:n, :nrhs], upper=False)

    def gemm_nn_sub(self, M, N, K, A, B, C):
        C[:M, :N] -= A[:M, :K] @ B[:K, :N]

    def sync(self):
        pass
