"""Row-sharded exact GP over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference's only parallel decomposition is gp2Scale's block decomposition of the covariance
over Dask workers (fvgp/gp2Scale_covariance.py:381-396, one broadcast of x: gp_prior.py:319-322).
Here the same pattern carries a DENSE factorisation:

  * x is replicated; 128-row blocks of K+V are dealt block-cyclically (block b -> rank b mod P),
    so every rank assembles exactly its own rows, in place, with no gather (contiguous row
    strips as gp2Scale's `ranges()` would leave the last rank 33 % of the flops);
  * right-looking blocked Cholesky with panel width NB, sequenced INSIDE the library
    (fvgp_hip_loglik_dist, fvgp_amd/csrc/dist_driver.h): per panel the NB x NB diagonal block is
    all-gathered from its owners, stacked on the rank's own rows of the panel and factored like a
    panel of the single-GPU driver; the panel factor is all-gathered (the one large collective,
    sum ~ 4 N^2 bytes per rank) and applied to the rank's block rows, with one panel of look-ahead
    (the chain of the next panel on the handle's high-priority stream under the trailing update);
    RCCL is called directly on that stream;
  * the forward solve rides along as one more block row holding (y-m)^T, replicated on every rank;
    log|KV| and (y-m)^T KV^-1 (y-m) come out replicated, so the log-likelihood needs no final
    reduction and one evaluation has exactly one host synchronisation.

This module owns the buffers (torch tensors as device-memory containers) and sequences what follows the
factorisation -- backward solve, posterior, gradient -- from ABI calls; every collective goes through
the handle (fvgp_hip_all_reduce / fvgp_hip_all_gather): RCCL when the process group's backend is
nccl, callbacks into torch.distributed otherwise (gloo: the CPU tests, which bind the CPU twin of
the ABI through tests/dist_stub_ops.py -- the analogue of the reference's in-process Dask cluster
fixture, tests/test_fvgp.py:20).
"""
import ctypes
import math

import numpy as np

from . import _lib

TILE = 128
DEFAULT_OPS_FACTORY = None        # callable() -> ops object used when ShardedGP is given none (None: HipOps); the CPU tests point it
                                  # at their host stand-in so that unpickled objects find it again


class HipOps:
    """The product implementation: every operation is a libfvgp_hip.so call on one handle (main stream = torch's
    current stream, the panel chain on the handle's own high-priority stream)."""

    def __init__(self, handle=None):
        from .device import default_handle
        self.torch = __import__("torch")
        self.H = handle or default_handle()
        self._stream = self.torch.cuda.current_stream(self.H.device)
        self.native_collectives = True          # fvgp_hip_comm_init (RCCL) is available

    def close(self):
        self.torch.cuda.synchronize(self.H.device)
        self.H.comm_destroy()

    def stream(self):
        """context in which torch's own work (copies) lands on this object's stream"""
        return self.torch.cuda.stream(self._stream)

    def zeros(self, *shape, dtype=None):
        if dtype is None:
            return self.H.zeros(*shape)
        return self.torch.zeros(*shape, dtype=dtype, device=f"cuda:{self.H.device}")

    def to_device(self, a):
        return self.H.to_device(a)

    def wrap(self, ptr, count):
        """fp64 tensor view of `count` doubles at the raw device pointer `ptr` (collective callbacks)"""
        class _Mem:
            __cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(_Mem(), device=f"cuda:{self.H.device}")

    def host_sync(self):
        self.torch.cuda.synchronize(self.H.device)

    # -- ABI pass-throughs -----------------------------------------------------------------------------------------
    def kmat(self, kernel_id, x1, x2, theta, out, vdiag=None, pad=_lib.PAD_ZERO):
        self.H.kmat(kernel_id, x1, x2, theta, out, vdiag=vdiag, pad=pad)

    def add_matrix(self, A, B, alpha=1.0):
        self.H.add_matrix(A, B, alpha)

    def dist_workspace(self, desc):
        return self.H.dist_workspace(desc)

    def loglik_dist(self, desc, theta):
        return self.H.loglik_dist(desc, theta)

    def dist_scratch(self, desc, what, npred=0, slab=TILE):
        return self.H.dist_scratch(desc, what, npred, slab)

    def solve_dist(self, desc, alpha_out, ws):
        self.H.solve_dist(desc, alpha_out, ws)

    def posterior_dist(self, desc, theta, xpred, npred, k_pre, kk_pre, alpha, mean_out, S_out, ws):
        self.H.posterior_dist(desc, theta, xpred, npred, k_pre, kk_pre, alpha, mean_out, S_out, ws)

    def grad_dist(self, desc, theta, alpha, component, slab, diag_out, ws):
        return self.H.grad_dist(desc, theta, alpha, component, slab, diag_out, ws)

    def all_reduce(self, t):
        self.H.all_reduce(t)

    def all_gather(self, send, recv):
        self.H.all_gather(send, recv)

    def comm_init(self, unique_id, rank, nranks):
        self.H.comm_init(unique_id, rank, nranks)

    def comm_init_callbacks(self, coll, rank, nranks):
        self.H.comm_init_callbacks(coll, rank, nranks)

    def ipc_window(self, window_bytes):
        return self.H.ipc_window(window_bytes)

    def comm_init_ipc(self, all_handles, shm_name, rank, nranks):
        self.H.comm_init_ipc(all_handles, shm_name, rank, nranks)

    def comm_profile(self):
        return self.H.comm_profile()

    def comm_info(self):
        return self.H.comm_info()

    def comm_check(self):
        self.H.comm_check()

    def set_option(self, key, value):
        self.H.set_option(key, value)

    def get_profile(self):
        return self.H.get_profile()

    def sync(self):
        self.H.sync()


def torch_collectives(ops, dist, group, world):
    """fvgp_collectives whose two entries run torch.distributed on views of the raw buffers -- the binding for process groups
    without RCCL (gloo).  Device buffers travel through the host (the collective is then synchronous: test-sized problems)."""
    torch = ops.torch

    def all_gather(ctx, send, recv, count, stream):
        try:
            s, r = ops.wrap(send, count), ops.wrap(recv, count * world)
            if s.is_cuda:
                ops.host_sync()
                sc = s.cpu()
                out = [torch.empty_like(sc) for _ in range(world)]
                dist.all_gather(out, sc, group=group)
                r.copy_(torch.cat(out))
                ops.host_sync()
            else:
                dist.all_gather(list(r.view(world, count).unbind(0)), s.clone(), group=group)
            return 0
        except Exception as e:                                 # noqa: BLE001 -- an exception cannot cross the C frame
            print(f"fvgp_amd.dist: all_gather callback failed: {type(e).__name__}: {e}", flush=True)
            return 2999

    def all_reduce(ctx, buf, count, stream):
        try:
            b = ops.wrap(buf, count)
            if b.is_cuda:
                ops.host_sync()
                bc = b.cpu()
                dist.all_reduce(bc, op=dist.ReduceOp.SUM, group=group)
                b.copy_(bc)
                ops.host_sync()
            else:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:                                 # noqa: BLE001
            print(f"fvgp_amd.dist: all_reduce callback failed: {type(e).__name__}: {e}", flush=True)
            return 2999

    return _lib.Collectives(None, _lib.ALL_GATHER_FN(all_gather), _lib.ALL_REDUCE_FN(all_reduce))


class ShardedGP:
    """log marginal likelihood of one GP sharded over the process group.

    x (n,d), y (n,) or (n,c), noise variances (n,) are given replicated (host arrays); the N x N
    matrix only ever exists as this rank's block rows.  Below them every rank keeps one more 128-row
    block holding (y-m)^T: carried through the panel solves and trailing updates like any other block
    row it comes out as (L^-1 (y-m))^T, so the forward solve costs no extra pass and no collective.

    force_collectives: a single rank still takes the panel-buffer path and calls its collectives (through RCCL when
    `collectives` is "rccl"): the whole multi-rank code path on one GPU."""

    def __init__(self, x, y, noise_variances, kernel="rbf_ard", group=None, ops=None, panel=1024,
                 rank=None, world=None, collectives="auto", force_collectives=False, ipc_window_bytes=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        in_group = dist.is_initialized() and rank is None
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.p, self.P = int(rank), int(world)
        assert panel % TILE == 0 and panel >= TILE, "panel width must be a multiple of 128"
        self.NB = int(panel)
        self.ops = ops if ops is not None else (DEFAULT_OPS_FACTORY() if DEFAULT_OPS_FACTORY is not None else HipOps())
        # a host callable k(x1, x2, theta) -> ndarray (gp_prior.py:217-224) is evaluated per rank for the rank's rows and uploaded
        # (slow path, N^2 / P over PCIe per evaluation); named kernels are assembled on the device
        self.kernel_callable = kernel if callable(kernel) else None
        self.kernel_id = 0 if self.kernel_callable is not None else (_lib.KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel))
        self.noise_matrix = None                                     # (n, n) host array: matrix-valued noise model (gp_kv.py:654-657)
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.x_host = x
        y = np.asarray(y, dtype=np.float64).reshape(len(x), -1)
        self.n, self.d = x.shape
        self.ncol = y.shape[1]
        assert self.ncol <= TILE, "at most 128 columns of y"
        self.np_ = _lib.pad128(self.n)
        self.nblk = self.np_ // TILE
        self.nb_max = -(-self.nblk // self.P)                       # block rows per rank (uniform, padded)
        self.nb_loc = len(range(self.p, self.nblk, self.P))          # block rows this rank really owns
        self.nloc = self.nb_max + 1                                  # + the block of right-hand-side rows
        self.general = self.P > 1 or bool(force_collectives)         # panel buffers + collectives (dist_driver.h)
        # global row index of every local row
        gb = np.arange(self.nb_max) * self.P + self.p
        self.gidx = (gb[:, None] * TILE + np.arange(TILE)[None, :]).reshape(-1)
        self.nv = int(np.sum(self.gidx < self.n))                    # valid (non-padding) local rows: a prefix
        assert np.all(self.gidx[:self.nv] < self.n)
        o = self.ops
        # ---- the collectives of this handle: RCCL (nccl process groups, or one forced rank) or torch.distributed callbacks
        user_coll = collectives if isinstance(collectives, _lib.Collectives) else None       # the caller's own (tools/shard_emulate.py)
        if user_coll is not None:
            collectives = "user"
        if collectives == "auto":
            nccl = in_group and dist.get_backend(group) == "nccl"
            collectives = "rccl" if (nccl or (self.P == 1 and force_collectives)) and getattr(o, "native_collectives", False) else "torch"
        self.collectives = collectives
        if self.general:
            if collectives == "rccl":
                uid = [_lib.comm_unique_id() if self.p == 0 else None]
                if self.P > 1:
                    dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                o.comm_init(uid[0], self.p, self.P)
            elif collectives == "ipc":
                # direct collectives over peer mappings (csrc/ipc.hip): every rank's window handle to every rank, the flag file's name
                # from rank 0; the bootstrap is the process group's own object collectives (any backend)
                import os
                import uuid
                window = int(ipc_window_bytes) if ipc_window_bytes else self._ipc_window_bytes()
                handle = o.ipc_window(window)
                if self.P > 1:
                    both = [None] * self.P
                    dist.all_gather_object(both, (handle, window), group=group)
                    handles = [b[0] for b in both]
                    # a peer's half b sits at b * (its window / 2): every rank must have asked for the same window
                    if len({b[1] for b in both}) != 1:
                        raise ValueError(f"ipc collectives: the ranks asked for different window sizes {[b[1] for b in both]}")
                    name = [f"/fvgp_ipc_{os.getpid()}_{uuid.uuid4().hex[:12]}" if self.p == 0 else None]
                    dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                else:
                    handles, name = [handle], [f"/fvgp_ipc_{os.getpid()}_{uuid.uuid4().hex[:12]}"]
                o.comm_init_ipc(handles, name[0], self.p, self.P)
                if self.P > 1:
                    dist.barrier(group=group)                       # every rank has mapped the flag file: its name can go
                if self.p == 0:
                    try:
                        os.unlink("/dev/shm" + name[0])
                    except OSError:
                        pass
            elif user_coll is not None:
                o.comm_init_callbacks(user_coll, self.p, self.P)
            elif self.P > 1:
                o.comm_init_callbacks(torch_collectives(o, dist, group, self.P), self.p, self.P)
        # ---- replicated inputs and this rank's buffers (sizes from the library)
        self.x_all = o.to_device(x)
        self.x_loc = o.to_device(x[self.gidx[:self.nv]]) if self.nv > 0 else None
        self.v_host = np.asarray(noise_variances, dtype=np.float64)
        self.v_dev = o.to_device(self.v_host)
        m = float(np.mean(y))                                        # default prior mean, gp_prior.py:449-458
        zt = np.zeros((TILE, self.np_))
        zt[:self.ncol, :self.n] = (y - m).T
        self.zt = o.to_device(zt)
        self.zrow = self.nb_max * TILE
        self.bnd = list(range(0, self.np_, self.NB)) + [self.np_]
        self.npan = len(self.bnd) - 1
        d = self._desc = _lib.DistDesc()
        d.n, d.d, d.ncol, d.panel, d.rank, d.nranks, d.kernel_id = self.n, self.d, self.ncol, self.NB, self.p, self.P, self.kernel_id
        d.force_general = 1 if (self.general and self.P == 1) else 0
        ws = o.dist_workspace(d)
        assert ws[0] == self.nloc * TILE * self.np_ and ws[5] == self.npan
        self.A = o.zeros(self.nloc * TILE, self.np_)
        self.info_dev = o.zeros(self.npan, dtype=torch.int32)
        self.ld_dev = o.zeros(self.npan)
        self._T = self._recv = [None, None]
        self._Dfac = self._gather = None
        if self.general:
            # tall panel: diagonal block + local rows; two of them, so that the trailing update of panel J can keep reading its
            # rows from the compact panel while the chain of panel J + 1 fills the other one; the factored NB x NB diagonal
            # blocks stay replicated for the later solves (N x NB doubles per rank, 0.4 GB at N = 50k)
            self._T = [o.zeros(ws[1]) for _ in range(2)]
            self._recv = [o.zeros(ws[2]) for _ in range(2)]
            self._Dfac = o.zeros(self.npan, self.NB, self.NB)
            self._gather = o.zeros(ws[4])
        d.x_all, d.vdiag, d.zt, d.A = self.x_all.data_ptr(), self.v_dev.data_ptr(), self.zt.data_ptr(), self.A.data_ptr()
        for i in range(2):
            d.T[i] = self._T[i].data_ptr() if self.general else None
            d.recv[i] = self._recv[i].data_ptr() if self.general else None
        d.Dfac = self._Dfac.data_ptr() if self.general else None
        d.gather = self._gather.data_ptr() if self.general else None
        d.info_dev, d.logdet_dev = self.info_dev.data_ptr(), self.ld_dev.data_ptr()
        self.keep_factor = True            # False: a likelihood-only evaluation leaves the factored panels out of A (no copy back)
        self.theta = None
        self.alpha = None                  # KVinvY, replicated, (np_, 128) with the first ncol columns in use

    def close(self):
        """Give the communicator back.  The ranks meet first (a group barrier after every rank's own work has drained): a rank
        that freed its IPC window or destroyed its RCCL communicator while a slower peer was still pulling from it would pull
        the rug from under that peer (the library also waits, bounded, for the peers' last pulls)."""
        o = self.ops
        if o is None:
            return
        if hasattr(o, "host_sync"):
            o.host_sync()
        if self.P > 1 and self.dist.is_initialized():
            try:
                self.dist.barrier(group=self.group)
            except Exception:                                   # noqa: BLE001 -- a dead peer must not keep this rank from cleaning up
                pass
        if hasattr(o, "close"):
            o.close()
        self.ops = None

    def comm_info(self):
        """the communicator as it reports itself ({"kind": "rccl", "nccl_comm_count": ...}); {} for ops without the query"""
        return self.ops.comm_info() if hasattr(self.ops, "comm_info") else {}

    def _ipc_window_bytes(self):
        """two halves, each large enough for the biggest all-gather piece of an evaluation (a rank's rows of a panel factor), capped at
        2 x 256 MB: larger calls (the gradient's all-reduces) are cut into pieces by the library"""
        piece = max(self.nloc * TILE * self.NB * 8, self.NB * self.NB * 8, 1 << 20)
        return 2 * min(piece, 256 << 20)

    # -- collectives of the parts sequenced here (through the handle: RCCL or the bound callbacks; nothing on one rank) ----
    def _all_reduce(self, t):
        if self.general:
            self.ops.all_reduce(t)

    def collective_summary(self):
        """{kind: (calls, bytes received per rank, milliseconds on their stream)} since the last call (option "profile")."""
        return self.ops.comm_profile() if hasattr(self.ops, "comm_profile") else {}

    def set_targets(self, ymean, noise_variances):
        """Replace (y - m) (n, ncol) and the noise variances (n,) -- the O(N) host-side results of the mean and noise
        functions at the hyperparameters about to be evaluated (gp_prior.py:226-234, gp_likelihood.py:89-110)."""
        ymean = np.asarray(ymean, dtype=np.float64).reshape(self.n, -1)
        assert ymean.shape[1] == self.ncol
        zt = np.zeros((TILE, self.np_))
        zt[:self.ncol, :self.n] = ymean.T
        self.zt.copy_(self.ops.to_device(zt))
        V = np.asarray(noise_variances, dtype=np.float64)
        if V.ndim == 2:                                              # K + V with a full matrix: its rows are added at assembly
            self.noise_matrix, self.v_host = V, np.zeros(self.n)
        else:
            self.noise_matrix, self.v_host = None, V
        self.v_dev.copy_(self.ops.to_device(self.v_host))

    def _kernel_rows(self, x2_host, x2_dev, theta, out):
        """out[:nv, :len(x2)] = k(this rank's points, x2) (padded window zeroed): device assembly, or the host callable"""
        o = self.ops
        if self.kernel_callable is None:
            if self.nv > 0:
                o.kmat(self.kernel_id, self.x_loc, x2_dev, theta, out)
            return
        out.zero_()
        if self.nv > 0:
            k = np.ascontiguousarray(self.kernel_callable(self.x_host[self.gidx[:self.nv]], x2_host, theta), dtype=np.float64)
            out[:self.nv, :k.shape[1]].copy_(o.to_device(k))

    def _assemble_rows(self, theta):
        """The rank's rows of K (+ its rows of a matrix-valued noise model) placed in A by this module: what the library's own
        assembly cannot do (Python kernel callables, 2-d noise).  The driver then adds the diagonal noise / identity padding."""
        o, A = self.ops, self.A
        with o.stream():
            A[:self.zrow].zero_()
            self._kernel_rows(self.x_host, self.x_all, theta, A[:self.zrow])
            if self.noise_matrix is not None and self.nv > 0:
                o.add_matrix(A[:self.nv, :self.n], o.to_device(self.noise_matrix[self.gidx[:self.nv]]))
        self._desc.preassembled = 1

    def evaluate(self, theta, want_alpha=False, keep_factor=True):
        """One pass of the path on the sharded matrix (fvgp_hip_loglik_dist: assemble, factor, the forward solve riding
        along), optionally the backward solve.  Returns (log-likelihood, log|KV|, (y-m)^T KV^-1 (y-m) / ncol), replicated.
        keep_factor=False (likelihood only): the factored panels are not copied back into the matrix, so no solve,
        gradient or posterior can follow this evaluation."""
        self.keep_factor = bool(keep_factor or want_alpha)
        self._desc.keep_factor = 1 if self.keep_factor else 0
        self._desc.preassembled = 0
        if self.kernel_callable is not None or self.noise_matrix is not None:
            self._assemble_rows(np.asarray(theta, dtype=np.float64))
        with self.ops.stream():
            ll, logdet, quad, info = self.ops.loglik_dist(self._desc, np.asarray(theta, dtype=np.float64))
        if info != 0:
            self.theta = None
            raise np.linalg.LinAlgError(f"{info}-th leading minor of the array is not positive definite")
        self.theta = np.array(theta, dtype=np.float64) if self.keep_factor else None
        self.alpha = None
        if want_alpha:
            self.solve_backward()
        return ll, logdet, quad

    def log_likelihood(self, theta):
        """GPMarginalLikelihood.log_likelihood(theta) (gp_marginal_likelihood.py:137-179) on the sharded matrix.
        Returns (log-likelihood, log|KV|, (y-m)^T KV^-1 (y-m) / ncol), replicated on every rank."""
        return self.evaluate(theta, keep_factor=False)

    # -- after the factorisation: one library call per method (fvgp_hip_solve_dist / _posterior_dist / _grad_dist; the panel sweeps,
    #    products and collectives are C, fvgp_amd/csrc/dist_driver.h) -------------------------------------------------------------
    def solve_backward(self):
        """KVinvY = L^-T z (gp_kv.py:574-593, the second half of cho_solve), replicated on every rank: (np_, 128), the first
        ncol columns are the solution."""
        o = self.ops
        with o.stream():
            alpha = o.zeros(self.np_, TILE)
            ws = o.zeros(o.dist_scratch(self._desc, 0))
            o.solve_dist(self._desc, alpha, ws)
        self.alpha = alpha
        return alpha

    def posterior(self, x_pred, want_cov=True):
        """k^T KVinvY and kk - k^T KV^-1 k (gp_posterior.py:139-182,229-288) at the factored hyperparameters: every rank
        assembles its own rows of k(x_data, x_pred) (the library does; with a host kernel callable this method does and hands
        them over); the mean and V^T V (V = L^-1 k) are summed over the ranks.
        Returns host arrays (P_pred, ncol) and (P_pred, P_pred) or None, replicated."""
        assert self.theta is not None, "evaluate() first"
        o = self.ops
        if self.alpha is None:
            self.solve_backward()
        x_pred = np.ascontiguousarray(x_pred, dtype=np.float64)
        npred = len(x_pred)
        pp = _lib.pad128(npred)
        with o.stream():
            xp = o.to_device(x_pred)
            k_pre = kk_pre = None
            if self.kernel_callable is not None:
                k_pre = o.zeros(self.nb_max * TILE, pp)
                self._kernel_rows(x_pred, xp, self.theta, k_pre)
                if want_cov and self.p == 0:
                    kk_pre = o.zeros(pp, pp)
                    kk_pre[:npred, :npred].copy_(o.to_device(np.ascontiguousarray(self.kernel_callable(x_pred, x_pred, self.theta), dtype=np.float64)))
            mean = o.zeros(pp, TILE)
            S = o.zeros(pp, pp) if want_cov else None
            ws = o.zeros(o.dist_scratch(self._desc, 1, npred))
            o.posterior_dist(self._desc, self.theta, xp, npred, k_pre, kk_pre, self.alpha, mean, S, ws)
            o.sync()
        return mean[:npred, :self.ncol].cpu().numpy(), (None if S is None else S[:npred, :npred].cpu().numpy())

    def gather_rows(self, rows_local, ncols=None):
        """Host array (n, ncols) of a row-distributed matrix from this rank's rows (nb_max*128 x >= ncols, block-cyclic like
        the matrix itself): one all-gather, replicated result.  What the reference keeps as whole host arrays (`GP.K`,
        `kv.Chol_factor`, fvgp/gp.py:625-635) is materialised this way, on request only."""
        o = self.ops
        rows = self.nb_max * TILE
        ncols = self.np_ if ncols is None else ncols
        send = rows_local[:rows, :ncols].contiguous().view(-1)
        recv = o.zeros(self.P * send.numel())
        if self.general:
            o.all_gather(send, recv)
        else:
            recv.copy_(send)
        o.sync()
        full = recv.view(self.P, self.nb_max, TILE, ncols).cpu().numpy()              # [rank, local block, row, col]
        out = full.transpose(1, 0, 2, 3).reshape(self.nb_max * self.P * TILE, ncols)     # global block = local * P + rank
        return out[:self.n]

    def kernel_matrix(self, theta):
        """K(theta) (n, n) on the host, replicated: every rank assembles its rows, one all-gather"""
        o = self.ops
        with o.stream():
            k = o.zeros(self.nb_max * TILE, self.np_)
            self._kernel_rows(self.x_host, self.x_all, np.asarray(theta, dtype=np.float64), k)
            return self.gather_rows(k)[:, :self.n].copy()

    def factor_matrix(self):
        """tril of the Cholesky factor (n, n) on the host, replicated (needs an evaluation that kept the factor)"""
        assert self.theta is not None, "evaluate(keep_factor=True) first"
        with self.ops.stream():
            return np.tril(self.gather_rows(self.A)[:, :self.n])

    def gradient(self, component=0, slab=2048, want_diag=False):
        """1/2 (tr(KV^-1 dK_i) - b^T dK_i b), b = KVinvY[:, component] (gp_marginal_likelihood.py:262-300) for the
        kernel-owned hyperparameters.  inv(L) is built by rows with the distributed forward solve (N^2 / P doubles per
        rank); the Gram matrix of the rank's rows (the sum over ranks is KV^-1, never formed) is walked in column slabs of
        `slab` columns -- an np_ x slab buffer, traced by the fused pass (fvgp_hip_grad_trace_cols) and dropped -- so the
        per-rank memory falls with the number of ranks; the (H,) partial results are summed over the ranks.
        want_diag: also return diag(KV^-1) (n,), replicated -- the row sums of squares of inv(L)'s columns, which the
        gradients of noise-function hyperparameters need (gp_marginal_likelihood.py:262-267)."""
        assert self.theta is not None, "evaluate() first"
        if self.kernel_callable is not None:
            raise NotImplementedError("the row-sharded gradient re-evaluates dK/dtheta inside its trace kernel: it takes the named kernels")
        o = self.ops
        if self.alpha is None:
            self.solve_backward()
        slab = max(TILE, (int(slab) // TILE) * TILE)
        with o.stream():
            dg = o.zeros(self.np_) if want_diag else None
            ws = o.zeros(o.dist_scratch(self._desc, 2, 0, slab))
            g = o.grad_dist(self._desc, self.theta, self.alpha, component, slab, dg, ws)
            o.sync()
        if not want_diag:
            return g
        return g, dg[:self.n].cpu().numpy()
