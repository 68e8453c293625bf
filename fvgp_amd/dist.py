"""Row-sharded exact GP over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference's only parallel decomposition is gp2Scale's block decomposition of the covariance
over Dask workers (fvgp/gp2Scale_covariance.py:381-396, one broadcast of x: gp_prior.py:319-322).
Here the same pattern carries a DENSE factorisation:

  * x is replicated; 128-row blocks of K+V are dealt block-cyclically (block b -> rank b mod P),
    so every rank assembles exactly its own rows, in place, with no gather (contiguous row
    strips as gp2Scale's `ranges()` would leave the last rank 33 % of the flops);
  * right-looking blocked Cholesky with panel width NB:
      1. the NB x NB diagonal block is summed to every rank (all_reduce of a zero-filled buffer,
         <= 8 MB) and stacked on top of the rank's own rows of the panel;
      2. that tall panel is factored like a panel of the single-GPU driver, 128 columns at a time
         (leaf, TRSM of every row below, in-panel update): the top block redundantly on every rank
         -- no pivot traffic inside the panel -- the rank's rows solved along the way;
      3. the panel factor is all-gathered (the one large collective: sum ~ 4 N^2 bytes per rank);
      4. each rank applies the trailing update to its own block rows (lower tiles only), reading
         the gathered panel in the order the all-gather left it (no re-ordering copy);
     with one panel of look-ahead: the update is split into the next panel's columns and the rest,
     and steps 1-3 of the next panel run on a second stream while the rest is applied;
  * the forward solve rides along as one more block row holding (y-m)^T, replicated on every rank;
    log|KV| and (y-m)^T KV^-1 (y-m) come out replicated, so the log-likelihood needs no final
    reduction and one evaluation has exactly one host synchronisation.

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
    """The product implementation: every operation is a libfvgp_hip.so call.  `chain` is the same set of
    operations bound to a second, high-priority stream: the panel chain (diagonal-block factorisation, panel
    solve, all-gather) runs there while the main stream applies the previous panel to the trailing matrix."""

    def __init__(self, handle=None, reserve_cus=0, n_cus=256, _stream=None, chain_everywhere=True):
        """reserve_cus > 0 (a multiple of 8): the chain gets that many compute units of its own and the main
        stream the rest, through CU-masked streams -- worth it once the panel chain, not the trailing update,
        is the critical path (many ranks, small local matrices).  Mask bit i is CU i/8 of XCD i%8 on this part
        (tools/cu_mask_probe.hip), so the last reserve_cus bits take reserve_cus/8 CUs from every XCD and
        both streams keep all eight L2s."""
        from .device import default_handle, local_device
        torch = self.torch = __import__("torch")
        if _stream is not None:                                    # the chain-side twin
            self.H, self._stream, self.chain = handle, _stream, self
            return
        self._owned = []
        if reserve_cus > 0:
            dev = handle.device if handle is not None else local_device()
            assert reserve_cus % 8 == 0 and 0 < reserve_cus < n_cus
            side_cus = list(range(n_cus - reserve_cus, n_cus))
            main_cus = sorted(set(range(n_cus)) - set(side_cus))
            sm = _lib.create_stream(dev, cu_mask=main_cus)
            self._owned = [sm]
            self._stream = torch.cuda.ExternalStream(sm, device=dev)
            if chain_everywhere:       # the chain may use every CU: the reserved ones are always free for it, the rest as they free up
                side = torch.cuda.Stream(device=dev, priority=-1)
            else:
                ss = _lib.create_stream(dev, cu_mask=side_cus)
                self._owned.append(ss)
                side = torch.cuda.ExternalStream(ss, device=dev)
            self.H = _lib.Handle(dev, stream=sm)
        else:
            self.H = handle or default_handle()
            self._stream = torch.cuda.current_stream(self.H.device)
            side = torch.cuda.Stream(device=self.H.device, priority=-1)
        self.chain = HipOps(_lib.Handle(self.H.device, stream=side.cuda_stream), _stream=side)

    def close(self):
        """Release what this object created: the chain-side handle, and with reserve_cus the main handle and both
        CU-masked streams -- handles first, they synchronise their stream when destroyed."""
        self.torch.cuda.synchronize(self.H.device)
        if self.chain is not self:
            self.chain.H.close()
        owned = getattr(self, "_owned", [])
        if owned:
            self.H.close()
        for st in owned:
            _lib.destroy_stream(st)
        self._owned = []

    def stream(self):
        """context in which torch's own work (copies, collectives) lands on this object's stream"""
        return self.torch.cuda.stream(self._stream)

    def fork(self):
        """the chain stream waits for everything enqueued on the main stream so far"""
        ev = self.torch.cuda.Event()
        ev.record(self._stream)
        self.chain._stream.wait_event(ev)

    def join(self):
        """the main stream waits for everything enqueued on the chain stream so far"""
        ev = self.torch.cuda.Event()
        ev.record(self.chain._stream)
        self._stream.wait_event(ev)

    def timestamp(self):
        """a timing event recorded on this object's stream (for the collective timings bench.py reports)"""
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self._stream)
        return ev

    def zeros(self, *shape, dtype=None):
        if dtype is None:
            return self.H.zeros(*shape)
        return self.torch.zeros(*shape, dtype=dtype, device=f"cuda:{self.H.device}")

    def to_device(self, a):
        return self.H.to_device(a)

    def kmat_rows(self, kernel_id, x_rows, x_all, theta, out):
        """out[:len(x_rows), :len(x_all)] = k(x_rows, x_all); the rest of the padded window is zeroed."""
        self.H.kmat(kernel_id, x_rows, x_all, theta, out, pad=_lib.PAD_ZERO)

    def panel_potrf_dev(self, T, w, rows, n_valid, info_dev, logdet_dev):
        self.H.panel_potrf_dev(T, w, rows, n_valid, info_dev, logdet_dev)

    def syrk_rowshard(self, M, N, K, A, B, C, scale, off, b_ranks, b_blocks, b_off):
        self.H.syrk_rowshard(M, N, K, A, B, C, scale, off, b_ranks, b_blocks, b_off)

    def kmat(self, kernel_id, x1, x2, theta, out, vdiag=None, pad=_lib.PAD_ZERO):
        self.H.kmat(kernel_id, x1, x2, theta, out, vdiag=vdiag, pad=pad)

    def gemm(self, a_kmajor, b_nmajor, lower, M, N, K, alpha, A, B, beta, C):
        self.H.gemm(a_kmajor, b_nmajor, lower, M, N, K, alpha, A, B, beta, C)

    def trsm_lower(self, L, n, B, nrhs):
        """B <- L^-1 B with a factored diagonal block this object keeps across evaluations: the handle's cached block
        inverses are keyed on the address, so they are dropped first"""
        self.H.invalidate_factor()
        self.H.trsm_lower(L, n, B, nrhs)

    def trsm_lower_t(self, L, n, B, nrhs):
        self.H.invalidate_factor()
        self.H.trsm_lower_t(L, n, B, nrhs)

    def grad_trace(self, kernel_id, x, theta, W, b, partial):
        return self.H.grad_trace(kernel_id, x, theta, W, b, partial)

    def sync(self):
        self.H.sync()


class ShardedGP:
    """log marginal likelihood of one GP sharded over the process group.

    x (n,d), y (n,) or (n,c), noise variances (n,) are given replicated (host arrays); the N x N
    matrix only ever exists as this rank's block rows.  Below them every rank keeps one more 128-row
    block holding (y-m)^T: carried through the panel solves and trailing updates like any other block
    row it comes out as (L^-1 (y-m))^T, so the forward solve costs no extra pass and no collective."""

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
        assert self.ncol <= TILE, "at most 128 columns of y"
        self.np_ = _lib.pad128(self.n)
        self.nblk = self.np_ // TILE
        self.nb_max = -(-self.nblk // self.P)                       # block rows per rank (uniform, padded)
        self.nb_loc = len(range(self.p, self.nblk, self.P))          # block rows this rank really owns
        self.nloc = self.nb_max + 1                                  # + the block of right-hand-side rows
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
        zt = np.zeros((TILE, self.np_))
        zt[:self.ncol, :self.n] = (y - m).T
        self.zt = o.to_device(zt)
        self.A = o.zeros(self.nloc * TILE, self.np_)
        self.zrow = self.nb_max * TILE
        # diagonal of the local rows: + noise on real rows, 1 on the padding rows of the last block
        inside = self.gidx < self.np_
        sel = self.gidx[inside]
        dv = np.ones(len(sel))
        dv[sel < self.n] = self.v_host[sel[sel < self.n]]
        dev = self.A.device
        self._diag_rows = torch.as_tensor(np.nonzero(inside)[0], device=dev)
        self._diag_cols = torch.as_tensor(sel, device=dev)
        self._diag_add = torch.as_tensor(dv, device=dev)
        self._diag_real = torch.as_tensor(sel < self.n, device=dev)
        # panels
        self.bnd = list(range(0, self.np_, self.NB)) + [self.np_]
        self.npan = len(self.bnd) - 1
        self.info_dev = o.zeros(self.npan, dtype=torch.int32)
        self.ld_dev = o.zeros(self.npan)
        if self.P > 1:
            # tall panel: diagonal block + local rows; two of them, so that the trailing update of panel J can keep reading its
            # rows from the compact panel while the chain of panel J + 1 fills the other one
            self._T = [o.zeros((self.NB + self.nloc * TILE) * self.NB) for _ in range(2)]
            self._low = [None, None]
            self._recv = [o.zeros(self.P * self.nb_max * TILE * self.NB) for _ in range(2)]
            # the factored NB x NB diagonal blocks, replicated: the panel-local part of every later solve
            # (N x NB doubles per rank, 0.4 GB at N = 50k)
            self._Dfac = o.zeros(self.npan, self.NB, self.NB)
        self.keep_factor = True            # False: a likelihood-only evaluation leaves the factored panels out of A (no copy back)
        self.theta = None
        self.alpha = None                  # KVinvY, replicated, (np_, 128) with the first ncol columns in use
        self._into_tensor = self.P > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.collective_events = None      # set to [] to collect (kind, bytes, start, end) per collective

    # -- collectives (no-ops on one rank) ---------------------------------------------------------
    def _all_reduce(self, t):
        if self.P > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _all_gather(self, out, inp):
        """out (P, k) <- every rank's inp (k,)"""
        if self._into_tensor:
            self.dist.all_gather_into_tensor(out.view(-1), inp, group=self.group)
        else:
            self.dist.all_gather(list(out.unbind(0)), inp, group=self.group)      # contiguous views (gloo)

    def _timed(self, kind, nbytes, fn, *a):
        rec = self.collective_events
        if rec is None or not hasattr(self.ops.chain, "timestamp"):
            return fn(*a)
        t0 = self.ops.chain.timestamp()
        fn(*a)
        rec.append((kind, nbytes, t0, self.ops.chain.timestamp()))

    def collective_summary(self):
        """{kind: (calls, bytes received per rank, milliseconds on the chain stream)} of the events collected so far."""
        self.ops.sync()
        self.ops.chain.sync()
        out = {}
        for kind, nbytes, t0, t1 in self.collective_events or []:
            c, b, ms = out.get(kind, (0, 0.0, 0.0))
            out[kind] = (c + 1, b + nbytes, ms + t0.elapsed_time(t1))
        return out

    # -- steps ------------------------------------------------------------------------------------
    def assemble(self, theta):
        """This rank's block rows of K+V (full width: the rows are short enough that skipping the upper
        part is not worth a second code path), identity on the padding, (y-m)^T in the extra block."""
        o, A = self.ops, self.A
        top = 0
        if self.nv > 0:
            o.kmat_rows(self.kernel_id, self.x_loc, self.x_all, np.asarray(theta, dtype=np.float64), A)
            top = _lib.pad128(self.nv)
        if top < self.zrow:
            A[top:self.zrow].zero_()
        rows, cols = self._diag_rows, self._diag_cols
        A[rows, cols] = self.torch.where(self._diag_real, A[rows, cols] + self._diag_add, self._diag_add)
        A[self.zrow:].copy_(self.zt)

    def _chain(self, J):
        """Panel J on the chain stream.  The diagonal block goes to every rank (all_reduce of a zero-filled
        buffer) and is stacked on top of this rank's rows of the panel; the tall panel is factored like a panel
        of the single-GPU driver (the top block redundantly on every rank -- no pivot traffic inside the panel);
        the solved rows are all-gathered."""
        o, A, P, p = self.ops.chain, self.A, self.P, self.p
        J0, Jend = self.bnd[J], self.bnd[J + 1]
        w = Jend - J0
        b0, b1 = J0 // TILE, Jend // TILE
        n_valid = max(0, min(w, self.n - J0))
        info, ld = self.info_dev[J:J + 1], self.ld_dev[J:J + 1]
        with o.stream():
            if P == 1:                                               # the panel is contiguous in A: in place
                o.panel_potrf_dev(A[J0:, J0:Jend], w, self.nloc * TILE - J0, n_valid, info, ld)
                return
            la = max(0, -(-(b0 - p) // P))                          # local blocks [la, lb) lie in the panel's rows
            lb = max(0, -(-(b1 - p) // P))
            L0 = b1 // P                                            # uniform first gathered local block (L0 <= lb)
            kt = (self.nloc - L0) * TILE                            # rows below: local blocks L0.. and the (y-m)^T block
            T = self._T[J % 2][:(w + kt) * w].view(w + kt, w)
            D, low = T[:w], T[w:]
            D.zero_()
            mine = None
            if lb > la:
                mine = D.view(w // TILE, TILE, w)[la * P + p - b0::P][:lb - la]
                mine.copy_(A[la * TILE:lb * TILE, J0:Jend].unflatten(0, (lb - la, TILE)))
            self._timed("all_reduce", 8.0 * w * w, self._all_reduce, D)
            low.copy_(A[L0 * TILE:, J0:Jend])
            o.panel_potrf_dev(T, w, w + kt, n_valid, info, ld)
            self._Dfac[J, :w, :w].copy_(D)
            self._low[J % 2] = low[(lb - L0) * TILE:]                # this rank's rows below the panel, compact (ld = w)
            if self.keep_factor:                                    # the solves that follow read the factor from A
                if mine is not None:
                    A[la * TILE:lb * TILE, J0:Jend].unflatten(0, (lb - la, TILE)).copy_(mine)
                A[lb * TILE:, J0:Jend].copy_(self._low[J % 2])
            else:                                                   # only the (y-m)^T rows are read back at the end
                A[self.zrow:, J0:Jend].copy_(low[(self.nb_max - L0) * TILE:])
            if Jend < self.np_:
                k = (self.nb_max - L0) * TILE
                self._timed("all_gather", 8.0 * (P - 1) * k * w, self._all_gather,
                            self._recv[J % 2][:P * k * w].view(P, k * w), low[:k].reshape(-1))

    def _update(self, J, c0, c1):
        """Apply panel J to block columns [c0, c1) of this rank's rows below the panel (lower tiles only)."""
        if c1 <= c0:
            return
        o, A, P, p = self.ops, self.A, self.P, self.p
        J0, Jend = self.bnd[J], self.bnd[J + 1]
        w = Jend - J0
        b1 = Jend // TILE
        l0 = max(0, -(-(b1 - p) // P))                              # first local block row below the panel
        M = (self.nloc - l0) * TILE
        cb = c0 // TILE
        if P > 1:
            L0 = b1 // P
            k = (self.nb_max - L0) * TILE
            B = self._recv[J % 2][:P * k * w].view(P * k, w)
            b_blocks, b_off = self.nb_max - L0, cb - L0 * P
        else:
            B, b_blocks, b_off = A[c0:self.zrow, J0:Jend], 0, 0
        Arows = A[l0 * TILE:, J0:Jend] if P == 1 else self._low[J % 2]      # P > 1: the compact panel (same values, ld = w)
        o.syrk_rowshard(M, c1 - c0, w, Arows, B, A[l0 * TILE:, c0:], P, l0 * P + p - cb,
                        P, b_blocks, b_off)

    def factor(self):
        """Blocked right-looking Cholesky of the sharded matrix with one panel of look-ahead, all enqueued
        without a host round trip; the appended rows come out as (L^-1 (y-m))^T."""
        o, bnd = self.ops, self.bnd
        o.fork()
        self._chain(0)
        for J in range(self.npan - 1):
            o.join()                                                # panel J factored and gathered
            self._update(J, bnd[J + 1], bnd[J + 2])                 # next panel's columns first ...
            o.fork()
            self._chain(J + 1)                                      # ... so its chain overlaps the rest
            self._update(J, bnd[J + 2], self.np_)
        o.join()

    def set_targets(self, ymean, noise_variances):
        """Replace (y - m) (n, ncol) and the noise variances (n,) -- the O(N) host-side results of the mean and noise
        functions at the hyperparameters about to be evaluated (gp_prior.py:226-234, gp_likelihood.py:89-110)."""
        ymean = np.asarray(ymean, dtype=np.float64).reshape(self.n, -1)
        assert ymean.shape[1] == self.ncol
        zt = np.zeros((TILE, self.np_))
        zt[:self.ncol, :self.n] = ymean.T
        self.zt.copy_(self.ops.to_device(zt))
        self.v_host = np.asarray(noise_variances, dtype=np.float64)
        sel = self._diag_cols.cpu().numpy()
        dv = np.ones(len(sel))
        dv[sel < self.n] = self.v_host[sel[sel < self.n]]
        self._diag_add.copy_(self.ops.to_device(dv))

    def _diag_block(self, J):
        """the factored diagonal block of panel J (lower triangle), on this rank"""
        J0, Jend = self.bnd[J], self.bnd[J + 1]
        if self.P == 1:
            return self.A[J0:Jend, J0:Jend]
        return self._Dfac[J, :Jend - J0, :Jend - J0]

    def _panel_rows(self, J):
        """(la, lb, first position, J0, Jend): local blocks [la, lb) are this rank's rows of panel J; they sit at the
        positions first, first + P, ... of the panel's 128-row blocks"""
        J0, Jend = self.bnd[J], self.bnd[J + 1]
        b0, b1 = J0 // TILE, Jend // TILE
        la = max(0, -(-(b0 - self.p) // self.P))
        lb = max(la, max(0, -(-(b1 - self.p) // self.P)))
        return la, lb, la * self.P + self.p - b0, J0, Jend

    def evaluate(self, theta, want_alpha=False, keep_factor=True):
        """One pass of the path on the sharded matrix: assemble, factor (the forward solve rides along), optionally
        the backward solve.  Returns (log-likelihood, log|KV|, (y-m)^T KV^-1 (y-m) / ncol), replicated.
        keep_factor=False (likelihood only): the factored panels are not copied back into the matrix, so no solve,
        gradient or posterior can follow this evaluation."""
        torch = self.torch
        self.keep_factor = bool(keep_factor or want_alpha)
        with self.ops.stream():
            self.assemble(theta)
            self.factor()
            z = self.A[self.zrow:self.zrow + self.ncol, :self.n]
            out = torch.cat([(z * z).sum().reshape(1), self.ld_dev.sum().reshape(1), self.info_dev.to(torch.float64)]).cpu().numpy()
        bad = np.nonzero(out[2:])[0]
        if len(bad):
            J = int(bad[0])
            self.theta = None
            raise np.linalg.LinAlgError(f"{self.bnd[J] + int(out[2 + J])}-th leading minor of the array is not positive definite")
        self.theta = np.array(theta, dtype=np.float64) if self.keep_factor else None
        self.alpha = None
        quad, logdet = float(out[0]) / self.ncol, float(out[1])
        if want_alpha:
            self.solve_backward()
        return -0.5 * (quad + logdet + self.n * math.log(2.0 * math.pi)), logdet, quad

    def log_likelihood(self, theta):
        """GPMarginalLikelihood.log_likelihood(theta) (gp_marginal_likelihood.py:137-179) on the sharded matrix.
        Returns (log-likelihood, log|KV|, (y-m)^T KV^-1 (y-m) / ncol), replicated on every rank."""
        return self.evaluate(theta, keep_factor=False)

    # -- solves with the distributed factor ---------------------------------------------------------
    def solve_backward(self):
        """KVinvY = L^-T z (gp_kv.py:574-593, the second half of cho_solve), replicated on every rank.
        Column sweep over the panels from the last to the first: the rows of panel J are solved against the replicated
        diagonal block (redundantly, no traffic), then every rank adds L[its rows of J, columns left of J]^T alpha_J to
        its own partial sum; one all-reduce of an NB x 128 slice per panel completes the right-hand side of the next."""
        o, A, P = self.ops, self.A, self.P
        with o.stream():
            Y = A[self.zrow:self.zrow + TILE, :self.np_].t().contiguous()     # z (np_, 128), replicated
            S = o.zeros(self.np_, TILE)                                       # this rank's partial sums
            alpha = o.zeros(self.np_, TILE)
            for J in range(self.npan - 1, -1, -1):
                la, lb, first, J0, Jend = self._panel_rows(J)
                w = Jend - J0
                G = S[J0:Jend].clone()
                self._all_reduce(G)
                G = Y[J0:Jend] - G
                o.trsm_lower_t(self._diag_block(J), w, G, TILE)
                alpha[J0:Jend].copy_(G)
                if lb > la and J0 > 0:
                    mine = G.view(w // TILE, TILE, TILE)[first::P][:lb - la].reshape(-1, TILE).contiguous()
                    o.gemm(1, 1, 0, J0, TILE, (lb - la) * TILE, 1.0, A[la * TILE:lb * TILE, :J0], mine, 1.0, S[:J0])
        self.alpha = alpha
        return alpha

    def forward_trsm(self, B, triangular=False):
        """B <- this rank's rows of L^-1 B_global, for a right-hand side distributed by rows like the matrix itself
        (B: nb_max*128 local rows x m columns, m a multiple of 128).  Per panel: the panel's rows are summed to every
        rank, solved against the replicated diagonal block, and applied to the rank's later rows as one GEMM.
        triangular: B_global is lower triangular (the identity: inv(L)), so panel J only carries its first Jend columns."""
        o, A, P = self.ops, self.A, self.P
        m = B.shape[1]
        assert m % TILE == 0 and B.shape[0] >= self.nb_max * TILE
        with o.stream():
            G_all = o.zeros(self.NB * m)
            for J in range(self.npan):
                la, lb, first, J0, Jend = self._panel_rows(J)
                w = Jend - J0
                mJ = min(m, Jend) if triangular else m
                G = G_all[:w * mJ].view(w, mJ)                           # contiguous: the collective needs it
                if P > 1:
                    G.zero_()
                    if lb > la:
                        G.unflatten(0, (w // TILE, TILE))[first::P][:lb - la].copy_(B[la * TILE:lb * TILE, :mJ].unflatten(0, (lb - la, TILE)))
                    self._all_reduce(G)
                else:
                    G.copy_(B[J0:Jend, :mJ])
                o.trsm_lower(self._diag_block(J), w, G, mJ)
                if lb > la:
                    B[la * TILE:lb * TILE, :mJ].unflatten(0, (lb - la, TILE)).copy_(G.unflatten(0, (w // TILE, TILE))[first::P][:lb - la])
                below = (self.nb_max - lb) * TILE
                if below > 0:
                    o.gemm(0, 1, 0, below, mJ, w, -1.0, A[lb * TILE:self.nb_max * TILE, J0:Jend], G, 1.0,
                           B[lb * TILE:self.nb_max * TILE, :mJ])
        return B

    def posterior(self, x_pred, want_cov=True):
        """k^T KVinvY and kk - k^T KV^-1 k (gp_posterior.py:139-182,229-288) at the factored hyperparameters: every rank
        assembles its own rows of k(x_data, x_pred); the mean and V^T V (V = L^-1 k) are summed over the ranks.
        Returns host arrays (P_pred, ncol) and (P_pred, P_pred) or None, replicated."""
        assert self.theta is not None, "evaluate() first"
        o = self.ops
        if self.alpha is None:
            self.solve_backward()
        x_pred = np.ascontiguousarray(x_pred, dtype=np.float64)
        npred = len(x_pred)
        pp = _lib.pad128(npred)
        rows = self.nb_max * TILE
        with o.stream():
            xp = o.to_device(x_pred)
            k = o.zeros(rows, pp)
            if self.nv > 0:
                o.kmat(self.kernel_id, self.x_loc, xp, self.theta, k)
            a_loc = o.zeros(rows, TILE)
            inside = self.gidx < self.np_
            a_loc[:int(inside.sum())] = self.alpha[self.torch.as_tensor(self.gidx[inside], device=self.alpha.device)]
            mean = o.zeros(pp, TILE)
            o.gemm(1, 1, 0, pp, TILE, rows, 1.0, k, a_loc, 0.0, mean)
            self._all_reduce(mean)
            S = None
            if want_cov:
                self.forward_trsm(k)
                S = o.zeros(pp, pp)
                if self.p == 0:
                    o.kmat(self.kernel_id, xp, xp, self.theta, S)
                o.gemm(1, 1, 0, pp, pp, rows, -1.0, k, k, 1.0, S)
                self._all_reduce(S)
            o.sync()
        return mean[:npred, :self.ncol].cpu().numpy(), (None if S is None else S[:npred, :npred].cpu().numpy())

    def gradient(self, component=0):
        """1/2 (tr(KV^-1 dK_i) - b^T dK_i b), b = KVinvY[:, component] (gp_marginal_likelihood.py:262-300) for the
        kernel-owned hyperparameters.  inv(L) is built by rows with the distributed forward solve; each rank forms the
        Gram matrix of ITS rows (the sum over ranks is KV^-1, never formed) and runs the fused trace pass on it; the
        (H,) partial results are summed over the ranks."""
        assert self.theta is not None, "evaluate() first"
        o, torch = self.ops, self.torch
        if self.alpha is None:
            self.solve_backward()
        rows = self.nb_max * TILE
        with o.stream():
            W = o.zeros(rows, self.np_)
            inside = self.gidx < self.np_
            li = torch.as_tensor(np.nonzero(inside)[0], device=W.device)
            W[li, torch.as_tensor(self.gidx[inside], device=W.device)] = 1.0          # this rank's rows of the identity
            self.forward_trsm(W, triangular=True)
            Gp = o.zeros(self.np_, self.np_)
            o.gemm(1, 1, 1, self.np_, self.np_, rows, 1.0, W, W, 0.0, Gp)
            del W
            nt = self.np_ // TILE
            partial = o.zeros(nt * (nt + 1) // 2 * (self.d + 2))
            b = self.alpha[:, component] if self.p == 0 else None
            g = o.grad_trace(self.kernel_id, self.x_all, self.theta, Gp, b, partial)
            gt = torch.as_tensor(g, device=Gp.device)
            self._all_reduce(gt)
            return gt.cpu().numpy()
