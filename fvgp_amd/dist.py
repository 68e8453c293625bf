"""Row-sharded exact GP over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference's only parallel decomposition is gp2Scale's block decomposition of the covariance
over Dask workers (fvgp/gp2Scale_covariance.py:381-396, one broadcast of x: gp_prior.py:319-322).
Here the same pattern carries a DENSE factorisation:

  * x is replicated; 128-row blocks of K+V are dealt block-cyclically (block b -> rank b mod P),
    so every rank assembles exactly its own rows, in place, with no gather (contiguous row
    strips as gp2Scale's `ranges()` would leave the last rank 33 % of the flops);
  * right-looking blocked Cholesky with panel width NB:
      1. the NB x NB diagonal block is summed to every rank (all_reduce of a zero-filled buffer,
         <= 8 MB) and factored redundantly -- no pivot traffic inside the panel;
      2. each rank solves its own rows of the panel against it (MFMA GEMMs);
      3. the panel factor is all-gathered (the one large collective: sum ~ 4 N^2 bytes per rank);
      4. each rank applies the trailing update to its own block rows (lower tiles only);
  * the forward solve is pipelined over panels with one small all_reduce per panel; log|KV| and
    (y-m)^T KV^-1 (y-m) come out replicated, so the log-likelihood needs no final reduction.

All arithmetic goes through an `ops` object; the product ops are the HIP kernels (HipOps, raises
without a GPU).  tests/ plug in a torch-CPU stand-in to exercise the partition and collective
logic under gloo -- the analogue of the reference's in-process Dask cluster fixture
(tests/test_fvgp.py:20).
"""
import math

import numpy as np

from . import _lib

TILE = 128


class HipOps:
    """The product implementation: every operation is a libfvgp_hip.so call."""

    def __init__(self, handle=None):
        from .device import default_handle
        self.H = handle or default_handle()
        self.torch = self.H.torch

    def zeros(self, *shape):
        return self.H.zeros(*shape)

    def to_device(self, a):
        return self.H.to_device(a)

    def kmat_rows(self, kernel_id, x_rows, x_all, theta, out):
        """out[:len(x_rows), :len(x_all)] = k(x_rows, x_all); the rest of the padded window is zeroed."""
        self.H.kmat(kernel_id, x_rows, x_all, theta, out, pad=_lib.PAD_ZERO)

    def potrf(self, D, n):
        return self.H.potrf(D, n)

    def panel_trsm(self, D, nd, Pm, rows):
        self.H.panel_trsm(D, nd, Pm, rows)

    def syrk_rowshard(self, M, N, K, A, B, C, scale, off):
        self.H.syrk_rowshard(M, N, K, A, B, C, scale, off)

    def trsm_lower(self, D, n, B, nrhs):
        self.H.trsm_lower(D, n, B, nrhs)

    def gemm_nn_sub(self, M, N, K, A, B, C):
        """C (M,N) -= A (M,K) @ B (K,N)."""
        self.H.gemm(0, 1, 0, M, N, K, -1.0, A, B, 1.0, C)

    def sync(self):
        self.H.sync()


class ShardedGP:
    """log marginal likelihood of one GP sharded over the process group.

    x (n,d), y (n,) or (n,c), noise variances (n,) are given replicated (host arrays); the N x N
    matrix only ever exists as this rank's block rows."""

    def __init__(self, x, y, noise_variances, kernel="rbf_ard", group=None, ops=None, panel=1024,
                 rank=None, world=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.p, self.P = int(rank), int(world)
        assert panel % TILE == 0 and panel >= TILE, "panel width must be a multiple of 128"
        self.NB = int(panel)
        self.ops = ops if ops is not None else HipOps()
        self.kernel_id = _lib.KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64).reshape(len(x), -1)
        self.n, self.d = x.shape
        self.ncol = y.shape[1]
        self.np_ = _lib.pad128(self.n)
        self.nblk = self.np_ // TILE
        self.nb_max = -(-self.nblk // self.P)                       # block rows per rank (uniform, padded)
        self.nb_loc = len(range(self.p, self.nblk, self.P))          # block rows this rank really owns
        # global row index of every local row
        gb = np.arange(self.nb_max) * self.P + self.p
        self.gidx = (gb[:, None] * TILE + np.arange(TILE)[None, :]).reshape(-1)
        self.nv = int(np.sum(self.gidx < self.n))                    # valid (non-padding) local rows: a prefix
        assert np.all(self.gidx[:self.nv] < self.n)
        o = self.ops
        self.x_all = o.to_device(x)
        self.x_loc = o.to_device(x[self.gidx[:self.nv]]) if self.nv > 0 else None
        self.v_host = np.asarray(noise_variances, dtype=np.float64)
        m = float(np.mean(y))                                        # default prior mean, gp_prior.py:449-458
        ym = np.zeros((self.np_, self.ncol))
        ym[:self.n] = y - m
        self.ymean_host = ym
        self.A = o.zeros(self.nb_max * TILE, self.np_)
        self._diag_rows = torch.as_tensor(np.arange(self.nb_max * TILE)[self.gidx < self.np_])
        self._diag_cols = torch.as_tensor(self.gidx[self.gidx < self.np_])
        dv = np.ones(len(self._diag_rows))                           # identity on the padding diagonal
        sel = self.gidx[self.gidx < self.np_]
        dv[sel < self.n] = self.v_host[sel[sel < self.n]]
        self._diag_add = dv

    # -- collectives (no-ops on one rank) ---------------------------------------------------------
    def _all_reduce(self, t):
        if self.P > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _all_gather(self, out, inp):
        if self.P > 1:
            chunks = list(out.view(self.P, -1).unbind(0))        # contiguous views: works on nccl and gloo
            self.dist.all_gather(chunks, inp.reshape(-1), group=self.group)
        else:
            out.copy_(inp.reshape(out.shape))

    # -- steps ------------------------------------------------------------------------------------
    def assemble(self, theta):
        """This rank's block rows of K+V (full width: the rows are short enough that skipping the upper
        part is not worth a second code path), identity on the padding."""
        o, A = self.ops, self.A
        A.zero_()
        if self.nv > 0:
            o.kmat_rows(self.kernel_id, self.x_loc, self.x_all, np.asarray(theta, dtype=np.float64), A)
        if self.nv < A.shape[0]:
            A[self.nv:].zero_()
        dev = A.device
        rows, cols = self._diag_rows.to(dev), self._diag_cols.to(dev)
        add = self.torch.as_tensor(self._diag_add, device=dev)
        valid = self.torch.as_tensor(self.gidx[self.gidx < self.np_] < self.n, device=dev)
        cur = A[rows, cols]
        A[rows, cols] = self.torch.where(valid, cur + add, add)

    def factor(self):
        """Blocked right-looking Cholesky of the sharded matrix, in place.  Returns (info, logdet) and keeps
        the factored diagonal blocks for the solves."""
        o, A, P, p, NB, np_ = self.ops, self.A, self.P, self.p, self.NB, self.np_
        torch = self.torch
        self.diag_blocks = []
        logdet = 0.0
        for J0 in range(0, np_, NB):
            Jend = min(J0 + NB, np_)
            w = Jend - J0
            b0, b1 = J0 // TILE, Jend // TILE
            # 1. diagonal block to everyone, factored redundantly
            D = o.zeros(w, w)
            for gb in range(b0, b1):
                if gb % P == p:
                    l = gb // P
                    D[(gb - b0) * TILE:(gb - b0 + 1) * TILE, :] = A[l * TILE:(l + 1) * TILE, J0:Jend]
            self._all_reduce(D)
            info = o.potrf(D, w)
            if info != 0:
                return J0 + info, float("nan")
            dg = torch.diagonal(D)[:max(0, min(w, self.n - J0))]
            logdet += 2.0 * float(torch.log(dg).sum().item())
            for gb in range(b0, b1):
                if gb % P == p:
                    l = gb // P
                    A[l * TILE:(l + 1) * TILE, J0:Jend] = D[(gb - b0) * TILE:(gb - b0 + 1) * TILE, :]
            self.diag_blocks.append(D)
            if Jend >= np_:
                break
            # 2. this rank's rows below the panel: X = A_panel * L_JJ^-T
            l0 = max(0, -(-(b1 - p) // P))                          # first local block row with global block >= b1
            rows = (self.nb_loc - l0) * TILE
            if rows > 0:
                o.panel_trsm(D, w, A[l0 * TILE:, J0:Jend], rows)
            # 3. all-gather the panel factor, re-ordered to global block order
            L0 = b1 // P                                            # uniform first local index on every rank
            m = self.nb_max - L0
            send = A[L0 * TILE:self.nb_max * TILE, J0:Jend].contiguous()
            recv = o.zeros(P * m * TILE, w)
            self._all_gather(recv, send)
            G = recv.view(P, m, TILE, w).permute(1, 0, 2, 3).reshape(m * P * TILE, w)
            Gv = G[(b1 - L0 * P) * TILE:]
            # 4. trailing update of this rank's block rows, lower tiles only
            N = np_ - Jend
            if rows > 0:
                if Gv.shape[0] < N:                                 # ranks past the end contribute nothing
                    pad = o.zeros(N - Gv.shape[0], w)
                    Gv = torch.cat([Gv, pad], dim=0)
                Gc = Gv[:N].contiguous()
                o.syrk_rowshard(rows, N, w, A[l0 * TILE:, J0:Jend], Gc, A[l0 * TILE:, Jend:], P, l0 * P + p - b1)
        return 0, logdet

    def forward_solve(self):
        """z = L^-1 (y - m), replicated; returns sum(z^2)/ncol.  One small all_reduce per panel."""
        o, A, P, p, NB, np_ = self.ops, self.A, self.P, self.p, self.NB, self.np_
        torch = self.torch
        c = self.ncol
        # local residual: rows this rank owns, 128 padded columns (GEMM granularity)
        r = o.zeros(self.nb_max * TILE, TILE)
        ym = o.to_device(self.ymean_host)
        rows_ok = self.gidx < np_
        r[torch.as_tensor(np.nonzero(rows_ok)[0], device=r.device), :c] = ym[torch.as_tensor(self.gidx[rows_ok], device=r.device)]
        quad = 0.0
        self.z_panels = []
        for Ji, J0 in enumerate(range(0, np_, NB)):
            Jend = min(J0 + NB, np_)
            w = Jend - J0
            b0, b1 = J0 // TILE, Jend // TILE
            D = self.diag_blocks[Ji]
            bJ = o.zeros(w, TILE)
            for gb in range(b0, b1):
                if gb % P == p:
                    l = gb // P
                    bJ[(gb - b0) * TILE:(gb - b0 + 1) * TILE] = r[l * TILE:(l + 1) * TILE]
            self._all_reduce(bJ)
            o.trsm_lower(D, w, bJ, TILE)                            # z_J = L_JJ^-1 b_J
            o.sync()
            quad += float((bJ[:, :c] ** 2).sum().item())
            self.z_panels.append(bJ)
            if Jend >= np_:
                break
            l0 = max(0, -(-(b1 - p) // P))
            rows = (self.nb_loc - l0) * TILE
            if rows > 0:
                o.gemm_nn_sub(rows, TILE, w, A[l0 * TILE:, J0:Jend], bJ, r[l0 * TILE:])
        return quad / c

    def log_likelihood(self, theta):
        """GPMarginalLikelihood.log_likelihood(theta) (gp_marginal_likelihood.py:137-179) on the sharded matrix."""
        self.assemble(theta)
        info, logdet = self.factor()
        if info != 0:
            raise np.linalg.LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        quad = self.forward_solve()
        return -0.5 * (quad + logdet + self.n * math.log(2.0 * math.pi)), logdet, quad
