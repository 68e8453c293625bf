"""GP -- the drop-in facade for the hot path of fvgp.GP (fvgp/gp.py:26,419-439).

Same constructor arguments, method names, argument meaning, return dictionaries and error
behaviour as the reference for the path BASELINE.json's north_star names:

    __init__ -> state build                         fvgp/gp.py:483-568, gp_kv.py:404-423
    log_likelihood / neg_log_likelihood             fvgp/gp.py:1310, gp_marginal_likelihood.py:137-179
    neg_log_likelihood_gradient                     fvgp/gp.py:1332, gp_marginal_likelihood.py:224-309
    posterior_mean / posterior_covariance           fvgp/gp.py:1376,1433, gp_posterior.py:139-182,229-288
    set_hyperparameters / update_gp_data            fvgp/gp.py:672-687,689
    properties K, V, m, hyperparameters, x_data, y_data   fvgp/gp.py:576-647

What differs, by design: x, y-m, V, the factor L and KVinvY live in HBM for the life of the
object; an evaluation moves theta in and a scalar / (H,) / (P,) / (P,P) result out.  All of
the arithmetic is in libfvgp_hip.so -- there is no CPU fallback (compute_device="cpu" raises).
The kernel is one of the named stationary kernels of fvgp_amd.kernels (None = the reference
default); any other callable is evaluated on the host exactly as the reference would and
uploaded (slow path, one N^2 transfer per evaluation, warned once).
"""
import inspect
import warnings

import numpy as np

from . import _lib
from . import kernels as _kernels
from .device import default_handle
from .gp_lin_alg import NonPositiveDefiniteError, _non_pd_message
from .gp_validation import ValidationMixin


class GP(ValidationMixin):
    def __init__(
        self,
        x_data,
        y_data,
        init_hyperparameters=None,
        noise_variances=None,
        compute_device="gpu",
        kernel_function=None,
        kernel_function_grad=None,
        noise_function=None,
        noise_function_grad=None,
        prior_mean_function=None,
        prior_mean_function_grad=None,
        gp2Scale=False,
        dask_client=None,
        gp2Scale_batch_size=10000,
        gp2Scale_distribution="blockwise",
        linalg_mode=None,
        ram_economy=False,
        args=None,
    ):
        # --- argument checks: fvgp/gp.py:441-452, gp_data.py:15-24 ---------------------------
        assert isinstance(noise_variances, np.ndarray) or noise_variances is None, "wrong format in noise_variances"
        assert init_hyperparameters is None or isinstance(init_hyperparameters, np.ndarray), "wrong init_hyperparameters"
        assert isinstance(compute_device, str), "wrong format in compute_device"
        assert (callable(kernel_function) or kernel_function is None
                or isinstance(kernel_function, str)), "wrong format in kernel_function"
        assert callable(noise_function) or noise_function is None, "wrong format in noise_function"
        assert callable(prior_mean_function) or prior_mean_function is None, "wrong format in prior_mean_function"
        assert isinstance(x_data, np.ndarray) and np.ndim(x_data) == 2, \
            "Euclidean x_data must be 2-d (n_points x input_dim); non-Euclidean inputs are not on the native path"
        assert isinstance(y_data, np.ndarray) and np.ndim(y_data) in (1, 2), "y_data must be a 1-d or 2-d np.ndarray"
        assert len(x_data) == len(y_data), "x_data and y_data do not have the same lengths."
        if compute_device not in ("gpu", "hip"):
            raise Exception("No valid compute device found. fvgp_amd runs on the MI355X only: "
                            "compute_device must be 'gpu' (there is no CPU path in this engine).")
        if gp2Scale:
            raise NotImplementedError("gp2Scale (sparse, Dask-distributed) is outside this engine's scope; "
                                      "the dense path scales by sharding over GPUs (fvgp_amd.dist).")
        # gp_kv.py:138-147: "Chol", "CholInv" / "Inv" (an explicit KV^-1 is kept next to the factor: the fast variance path of
        # gp_posterior.py:238-244), or three callables [f_factor, f_solve, f_logdet] that take over the linear algebra on HOST
        # arrays exactly as in the reference (gp_kv.py:457-458,552-554,625-628,697-698,715; documented gp.py:274-281)
        self._linalg_callables = None
        if isinstance(linalg_mode, (list, tuple)):
            assert len(linalg_mode) == 3 and all(callable(f) for f in linalg_mode), \
                "linalg_mode as a list needs three callables [f_factor(KV), f_solve(obj, b), f_logdet(obj)]"
            self._linalg_callables = list(linalg_mode)
            warnings.warn("linalg_mode callables work on host arrays: K+V crosses PCIe for every evaluation (slow path).", stacklevel=2)
            linalg_mode = "Chol"
        if linalg_mode not in (None, "Chol", "CholInv", "Inv"):
            raise NotImplementedError("the dense modes 'Chol', 'CholInv', 'Inv' or three callables run here; the sparse modes belong to gp2Scale")
        self.linalg_mode = {"Inv": "CholInv"}.get(linalg_mode, linalg_mode) or "Chol"
        self._KVinv = None
        self._custom_obj = None
        if isinstance(noise_variances, np.ndarray):
            assert np.ndim(noise_variances) == 1, "noise_variances must be 1-d"
            assert len(noise_variances) == len(y_data), "noise_variances and y_data have different lengths"
            assert np.all(noise_variances > 0.0), "all noise_variances must be positive"
        if noise_variances is not None and callable(noise_function):
            raise Exception("Noise function and measurement noise provided. Decide which one to use.")

        self.args = {} if args is None else args
        # posterior covariance at many points (gp_posterior.py:229-288): points per device chunk and the cap on the resident L^-1 k
        # scratch (None: a third of the free device memory), see _posterior_chunked
        self._posterior_chunk = int(self.args.get("posterior_chunk", 4096))
        self._posterior_scratch_bytes = self.args.get("posterior_scratch_bytes")
        self._posterior_groups = 0
        self.compute_device = "gpu"
        self.ram_economy = ram_economy
        # distribution switch (the reference's gp2Scale= / dask_client= constructor flags, gp.py:419-439):
        # args["process_group"] = a torch.distributed group (or True for the default group) row-shards K+V over
        # the group's ranks, one process per GPU (fvgp_amd/dist.py); every rank makes the same calls
        self._sharded = self.args.get("process_group") is not None and self.args.get("process_group") is not False
        self._sh = self._sh_work = None
        if self._sharded:
            # every linalg mode keeps the distributed Cholesky factor ('CholInv' caches an explicit inverse in the reference,
            # gp_kv.py:429-432: a speed choice with the same results); kernel callables are evaluated per rank for its rows
            self._H = None
        else:
            self._H = default_handle()
        self._set_data(x_data, y_data, noise_variances)
        self.x_out = None

        # --- kernel / mean / noise selection: gp_prior.py:57-93, gp_likelihood.py:27-38 -----------
        self._native = _kernels.resolve(kernel_function)
        self._kernel_callable = None if self._native is not None else kernel_function
        self._k_n_params = 3 if self._native is not None else len(inspect.signature(kernel_function).parameters)
        self._kernel_grad_callable = kernel_function_grad
        self._mean_callable = prior_mean_function
        self._m_n_params = 2 if prior_mean_function is None else len(inspect.signature(prior_mean_function).parameters)
        self._mean_grad_callable = prior_mean_function_grad
        self._noise_callable = noise_function
        self._v_n_params = 2 if noise_function is None else len(inspect.signature(noise_function).parameters)
        self._noise_grad_callable = noise_function_grad
        if noise_function is None and noise_variances is None:
            warnings.warn("No noise function or measurement noise provided. "
                          "Noise variances will be set to (0.01 * mean(|y_data|))^2.", stacklevel=2)
        if self._native is None:
            warnings.warn("kernel_function is a Python callable: it is evaluated on the host and the N x N "
                          "matrix is uploaded for every evaluation (slow path). Use a named kernel from "
                          "fvgp_amd.kernels to assemble on the GPU.", stacklevel=2)

        # --- hyperparameters: gp.py:488-505 ---------------------------------------------------
        user_callables = (self._native is None) or callable(prior_mean_function) or callable(noise_function)
        if init_hyperparameters is None:
            if user_callables:
                raise Exception("You have provided callables for kernel, mean, or noise functions but no "
                                "initial hyperparameters.")
            init_hyperparameters = np.ones(self._native.n_hyperparameters(self.index_set_dim))
            warnings.warn("Hyperparameters initialized to a vector of ones.")
        self._work = None      # scratch KV for evaluations at new theta (never the state)
        self._work2 = None
        self._alpha_work = None
        self._K_host = None
        self.set_hyperparameters(np.array(init_hyperparameters, dtype=np.float64))

    # ------------------------------------------------------------------------------------------
    # data
    # ------------------------------------------------------------------------------------------
    def _set_data(self, x_data, y_data, noise_variances):
        if np.ndim(y_data) == 1:                                  # gp_data.py:24
            y_data = y_data.reshape(len(y_data), 1)
        self.x_data = np.ascontiguousarray(x_data, dtype=np.float64)
        self.y_data = np.ascontiguousarray(y_data, dtype=np.float64)
        self.noise_variances = noise_variances
        self.index_set_dim = self.input_set_dim = self.x_data.shape[1]
        self.point_number = len(self.x_data)
        if self.index_set_dim > 16:
            raise NotImplementedError("the assembly kernels take input dimension <= 16")
        n = self.point_number
        if self._sharded:
            self._np = _lib.pad128(n)
            self._sh = self._sh_work = None          # built on first use (the kernel is resolved after this call)
            return
        H = self._H
        self._np = _lib.pad128(n)
        # the square buffers an evaluation factors in: one more block row where padded_dim(n) leaves no room for the appended (y-m)^T
        # (n a multiple of 128): the forward solve then rides in the factorisation for every n (fvgp_hip_loglik_dim)
        ncol = self.y_data.shape[1]
        self._ld = _lib.loglik_dim(n, ncol) if ncol <= _lib.MAX_RHS_VEC else self._np
        self._x_dev = H.to_device(self.x_data)
        self._L = H.empty(self._ld, self._ld)                     # state: factor of K+V at self.hyperparameters
        self._alpha = H.empty(self._np, self.y_data.shape[1])     # state: KVinvY
        self._work = self._work2 = self._alpha_work = None

    @property
    def hyperparameters(self):
        return self._hps

    # ------------------------------------------------------------------------------------------
    # mean / noise on the host (O(N)), exactly the reference's rules
    # ------------------------------------------------------------------------------------------
    def _mean(self, x, hps):
        """gp_prior.py:226-234,449-458: default = mean over ALL y entries, also at prediction points."""
        if self._mean_callable is None:
            m = np.zeros(len(x))
            m[:] = np.mean(self.y_data)
            return m
        m = self._mean_callable(x, hps) if self._m_n_params == 2 else self._mean_callable(x, hps, self.args)
        assert np.ndim(m) == 1, "mean function returned non-1-d result: " + str(m)
        return np.asarray(m, dtype=np.float64)

    def _noise(self, x, hps):
        """gp_likelihood.py:89-110."""
        if self._noise_callable is not None:
            v = self._noise_callable(x, hps) if self._v_n_params == 2 else self._noise_callable(x, hps, self.args)
            v = np.asarray(v, dtype=np.float64)
            assert np.ndim(v) in (1, 2), "V has strange dimensionality"          # gp_kv.py:653
            return v
        if self.noise_variances is not None:
            if len(x) == len(self.noise_variances):
                return self.noise_variances
            return np.zeros(len(x)) + np.mean(self.noise_variances)
        return np.ones(len(x)) * (np.mean(abs(self.y_data)) / 100.0) ** 2

    def _host_kernel(self, x1, x2, hps):
        k = self._kernel_callable(x1, x2, hps) if self._k_n_params == 3 else self._kernel_callable(x1, x2, hps, self.args)
        return np.ascontiguousarray(k, dtype=np.float64)

    # ------------------------------------------------------------------------------------------
    # one full pass of the hot path into (KV buffer, alpha buffer); returns (loglik, logdet, m, V)
    # ------------------------------------------------------------------------------------------
    def _make_sharded(self):
        from .dist import ShardedGP
        pg = self.args.get("process_group")
        return ShardedGP(self.x_data, self.y_data, np.ones(self.point_number),
                         kernel=self._native.kernel_id if self._native is not None else (lambda a, b, h: self._host_kernel(a, b, h)),
                         group=None if pg is True else pg, ops=self.args.get("shard_ops"),
                         panel=int(self.args.get("shard_panel", 1024)), rank=self.args.get("shard_rank"),
                         world=self.args.get("shard_world"))

    def _evaluate_sharded(self, hps, state, keep_factor=True):
        """The same pass on the row-sharded matrix: `state` True evaluates into the object that holds the GP's state
        (KVinvY included), False into the scratch twin (gp_kv.py:574-578: an evaluation at another theta touches no state)."""
        hps = np.asarray(hps, dtype=np.float64)
        if self._sh is None:
            self._sh = self._make_sharded()
        if not state and self._sh_work is None:
            self._sh_work = self._make_sharded()
        sh = self._sh if state else self._sh_work
        m = self._mean(self.x_data, hps)
        V = self._noise(self.x_data, hps)
        sh.set_targets(self.y_data - m[:, None], V)                    # 2-d V: the ranks add their rows of it (gp_kv.py:654-657)
        try:
            ll, logdet, _ = sh.evaluate(hps, want_alpha=state, keep_factor=keep_factor)
        except np.linalg.LinAlgError as e:
            raise NonPositiveDefiniteError(_non_pd_message(self.point_number, str(e).split("-th")[0],
                                                           float(np.min(V if np.ndim(V) == 1 else np.diag(V))), 0.0)) from e
        return ll, logdet, m, V, sh

    def _evaluate(self, hps, KV, alpha, need_alpha=True, use_callables=True):
        """need_alpha=False: the caller wants the likelihood only.  The forward solve rides along in the factorisation
        (quad = |L^-1 (y-m)|^2), so the backward solve that would produce KVinvY is skipped when the fused call can
        run without it (it needs ncol free padding rows)."""
        H, n = self._H, self.point_number
        hps = np.asarray(hps, dtype=np.float64)
        m = self._mean(self.x_data, hps)
        V = self._noise(self.x_data, hps)
        ymean = self.y_data - m[:, None]
        ym_dev = H.to_device(ymean)
        ncol = ymean.shape[1]
        # the fused call bounds the appended (y-m)^T rows with 1 + |y-m|^2 / min(V): it needs positive noise
        V2 = V if np.ndim(V) == 2 else None                        # matrix-valued noise: KV = K + V (gp_kv.py:654-657)
        if V2 is not None:
            V = np.ascontiguousarray(np.diag(V2))
        if self._linalg_callables is not None and use_callables:
            return self._evaluate_callables(hps, KV, alpha, m, V, V2, ymean)
        if V2 is None and self._native is not None and ncol <= _lib.MAX_RHS_VEC and float(np.min(V)) > 0.0:
            skip = (not need_alpha) and (KV.shape[0] - n) >= ncol
            ll, logdet, quad, info = H.loglik(self._native.kernel_id, self._x_dev, hps, H.to_device(V), ym_dev, KV,
                                              None if skip else alpha)
        else:
            if self._native is not None:
                H.kmat(self._native.kernel_id, self._x_dev, self._x_dev, hps, KV, vdiag=None if V2 is not None else H.to_device(V),
                       uplo=_lib.LOWER, pad=_lib.PAD_IDENTITY)
            else:
                K = self._host_kernel(self.x_data, self.x_data, hps)     # slow path: N^2 over PCIe
                if V2 is None:
                    K = K.copy()
                    np.fill_diagonal(K, np.diag(K) + V)                  # addKV on the host (gp_kv.py:665-667), then one upload
                KV[:n, :n] = H.to_device(K)
            if V2 is not None:
                H.add_lower(KV, n, H.to_device(V2))                      # one N^2 upload per evaluation, added on the device
            info = H.potrf(KV, n)
            ll = logdet = float("nan")
            if info == 0:
                rhs = alpha if ncol <= _lib.MAX_RHS_VEC else H.zeros(self._np, _lib.pad128(ncol))
                rhs[:n, :ncol] = ym_dev
                H.potrs(KV, n, rhs, ncol if ncol <= _lib.MAX_RHS_VEC else _lib.pad128(ncol))
                if rhs is not alpha:
                    alpha[:n] = rhs[:n, :ncol]
                logdet = H.logdet(KV, n)
                quad = H.dot(ym_dev, alpha, n) / ncol
                ll = -0.5 * (quad + logdet + n * np.log(2.0 * np.pi))
        if info != 0:
            raise NonPositiveDefiniteError(_non_pd_message(n, info, float(np.min(V)) if self._native is not None else None, 0.0))
        return ll, logdet, m, (V if V2 is None else V2)

    def _host_KV(self, hps, KV, V, V2):
        """K + V as a full symmetric host array (what GPkv.addKV hands to the callables, gp_kv.py:639-669)"""
        H, n = self._H, self.point_number
        if self._native is not None:
            H.kmat(self._native.kernel_id, self._x_dev, self._x_dev, hps, KV, vdiag=None if V2 is not None else H.to_device(V),
                   uplo=_lib.LOWER, pad=_lib.PAD_IDENTITY)
            H.symmetrize(KV, n)
            H.sync()
            K = KV[:n, :n].cpu().numpy().copy()
        else:
            K = self._host_kernel(self.x_data, self.x_data, hps)
            if V2 is None:
                K = K.copy()
                np.fill_diagonal(K, np.diag(K) + V)
        return K + V2 if V2 is not None else K

    def _evaluate_callables(self, hps, KV, alpha, m, V, V2, ymean):
        """compute_new_KVlogdet_KVinvY with user linear algebra (gp_kv.py:625-628): factor = f(KV), KVinvY = f_solve(factor,
        y-m).reshape(y.shape), logdet = f_logdet(factor); the factor object is kept with the buffer it was computed into."""
        f_factor, f_solve, f_logdet = self._linalg_callables
        n = self.point_number
        obj = f_factor(self._host_KV(hps, KV, V, V2))
        a = np.asarray(f_solve(obj, ymean), dtype=np.float64).reshape(ymean.shape)
        logdet = float(f_logdet(obj))
        alpha[:n] = self._H.to_device(a)
        if KV is self._L:
            self._custom_obj = obj
        else:
            self._custom_obj_work = obj
        ll = -0.5 * (np.sum(ymean * a) / ymean.shape[1] + logdet + n * np.log(2.0 * np.pi))
        return ll, logdet, m, (V if V2 is None else V2)

    def _eval_dim(self):
        """rows = columns of the square scratch an evaluation factors in (fvgp_hip_loglik_dim); the factor's own buffer may be larger
        (append headroom) -- the scratch never is"""
        ncol = self.y_data.shape[1]
        return _lib.loglik_dim(self.point_number, ncol) if ncol <= _lib.MAX_RHS_VEC else self._np

    def _scratch(self):
        if self._work is None:
            d = self._eval_dim()
            self._work = self._H.empty(d, d)
            self._alpha_work = self._H.empty(self._np, self.y_data.shape[1])
        return self._work, self._alpha_work

    # ------------------------------------------------------------------------------------------
    # state
    # ------------------------------------------------------------------------------------------
    def set_hyperparameters(self, hps):
        """fvgp/gp.py:672-687 -> prior, likelihood and KV state refresh (gp_kv.py:404-423)."""
        assert isinstance(hps, np.ndarray), "wrong format in hyperparameters"
        assert np.ndim(hps) == 1, "wrong format in hyperparameters"
        self._hps = np.array(hps, dtype=np.float64)
        if self._sharded:
            ll, logdet, m, V, _ = self._evaluate_sharded(self._hps, state=True)
        else:
            ll, logdet, m, V = self._evaluate(self._hps, self._L, self._alpha)
            if self._native is not None and self._linalg_callables is None:
                # the state changed: what posterior queries on it need is enqueued now (gp_kv.py:404-428 refreshes KVinvY and the
                # log-det the same way), not in front of the first query's sweep; the host does not wait for it
                self._H.posterior_prepare(self._L, self.point_number)
        self._loglik, self._logdet, self.m, self.V = ll, logdet, m, V
        self._K_host = None
        self._refresh_inverse()

    def _refresh_inverse(self):
        """CholInv mode (gp_kv.py:429-432): keep KV^-1 = POTRI(L), full symmetric, next to the factor."""
        if self.linalg_mode != "CholInv" or self._sharded:
            self._KVinv = None
            return
        H, n = self._H, self.point_number
        inv = self._L[:self._np, :self._np].clone()              # (the factor's buffer may carry append headroom: not cloned)
        H.invalidate_factor()                                     # `inv` may sit where a freed factor used to be
        work = H.empty(self._np, self._np)
        H.potri(inv, n, work)
        H.symmetrize(inv, n)
        if inv.shape[0] > n:
            inv[n:, :] = 0.0
            inv[:, n:] = 0.0
        self._KVinv = inv

    def get_hyperparameters(self):
        return self._hps

    def update_gp_data(self, x_new, y_new, noise_variances_new=None, append=True, rank_n_update=None):
        """fvgp/gp.py:689-779.  append=True extends the data; with rank_n_update (default = append) the factor is
        extended by bordering on the device -- v = L^-1 k(x_old, x_new), L22 = chol(K22 + V22 - v^T v)
        (cholesky_update_rank_n, gp_lin_alg.py:1310-1477; GPkv.update_KV, gp_kv.py:462-476) -- O(n^2 m) instead of
        the O(n^3) refactorisation; KVinvY and log|KV| are then refreshed from the new factor (gp_kv.py:404-423)."""
        assert isinstance(x_new, np.ndarray) and np.ndim(x_new) == 2, \
            "wrong format in x_new: a 2-d np.ndarray (non-Euclidean lists are not on the native path)"
        assert isinstance(y_new, np.ndarray) and np.ndim(y_new) in (1, 2), "wrong format in y_new"
        assert ((isinstance(noise_variances_new, np.ndarray) and np.ndim(noise_variances_new) == 1)
                or noise_variances_new is None), "noise_variances_new must be a 1-d np.ndarray or None"
        assert len(x_new) == len(y_new), "updated x and y do not have the same lengths."
        if rank_n_update is None:
            rank_n_update = append
        if not append and rank_n_update:                              # gp.py:733-737
            warnings.warn("`rank_n_update=True` is invalid when `append=False` (the previous factorization belongs "
                          "to data that no longer exists). Forcing `rank_n_update=False`.")
            rank_n_update = False
        if self.noise_variances is not None and noise_variances_new is None:          # gp_data.py:84-89
            raise Exception("Please provide noise_variances in the data update because you did at initialization "
                            "or during a previous update.")
        if self.noise_variances is None and noise_variances_new is not None:
            raise Exception("You did not initialize noise and but included noise in the update."
                            "Please reinitialize in this case.")
        if np.ndim(y_new) == 1:
            y_new = y_new.reshape(len(y_new), 1)
        if not append:
            self._set_data(x_new, y_new, noise_variances_new)
            self.set_hyperparameters(self._hps)
            return
        assert x_new.shape[1] == self.x_data.shape[1] and y_new.shape[1] == self.y_data.shape[1], \
            "appended data must have the column counts of the existing data"
        x = np.vstack([self.x_data, x_new])
        y = np.vstack([self.y_data, y_new])
        nv = None if self.noise_variances is None else np.concatenate([self.noise_variances, noise_variances_new])
        n_old = self.point_number
        if not (rank_n_update and self._native is not None and len(x_new) > 0) or self._sharded or self._linalg_callables is not None:
            self._set_data(x, y, nv)
            self.set_hyperparameters(self._hps)
            return
        self._append_factor(x, y, nv, n_old)

    def _append_factor(self, x, y, nv, n_old):
        """the bordered factor; all or nothing: a failure anywhere (a Schur complement that is not positive definite, a failed
        device call) leaves the object exactly as it was -- data, factor (its padding rows included) and cached results"""
        keep = {k: self.__dict__[k] for k in ("x_data", "y_data", "noise_variances", "point_number", "_np", "_ld", "_L", "_x_dev", "_alpha")}
        touched = []                      # [L_old] once its padding rows have been written
        try:
            self._append_factor_body(x, y, nv, n_old, touched)
        except BaseException:
            self.__dict__.update(keep)
            if touched:
                L_old, n = touched[0], len(x)
                L_old[n_old:n, :] = 0.0
                L_old[n_old:n, n_old:n].fill_diagonal_(1.0)           # identity padding again
                self._H.invalidate_factor()
            raise

    def _append_factor_body(self, x, y, nv, n_old, touched):
        H, hps, kid = self._H, self._hps, self._native.kernel_id
        L_old, np_old = self._L, self._np
        n, m = len(x), len(x) - n_old
        self.x_data = np.ascontiguousarray(x, dtype=np.float64)
        self.y_data = np.ascontiguousarray(y, dtype=np.float64)
        self.noise_variances = nv
        self.point_number = n
        V = self._noise(self.x_data, hps)
        mean = self._mean(self.x_data, hps)
        x_old_dev = self._x_dev
        x_new_dev = H.to_device(self.x_data[n_old:])
        mp, np_new = _lib.pad128(m), _lib.pad128(n)
        # v = L^-1 k(x_old, x_new)
        B = H.zeros(np_old, mp)
        H.kmat(kid, x_old_dev, x_new_dev, hps, B, pad=_lib.PAD_ZERO)
        H.trsm_lower(L_old, n_old, B, mp)
        # Schur complement of the new block and its factor
        S = H.zeros(mp, mp)
        H.kmat(kid, x_new_dev, x_new_dev, hps, S, vdiag=H.to_device(V[n_old:]), uplo=_lib.LOWER, pad=_lib.PAD_IDENTITY)
        H.gemm(1, 1, 0, mp, mp, np_old, -1.0, B, B, 1.0, S)
        info = H.potrf(S, m)
        if info != 0:
            raise NonPositiveDefiniteError(
                "Cholesky rank-n update failed: the Schur complement of the appended block is not positive definite "
                f"(leading minor {info}). This usually indicates the new data rows are linearly dependent on old rows "
                "or the kernel is not PD on the augmented set.")
        # the bordered factor [[L, 0], [v^T, L22]]: IN PLACE while the new rows fit into the padding rows of the buffer the factor
        # lives in (the appended rows only ever touch rows n_old .. n: autonomous-experiment loops append a few points per step,
        # gp_kv.py:462-476, and copying an N x N factor per step was a quarter of an append at N = 20k), else in a buffer of the
        # new size
        ncol = self.y_data.shape[1]
        ld_new = _lib.loglik_dim(n, ncol) if ncol <= _lib.MAX_RHS_VEC else np_new
        in_place = ld_new <= L_old.shape[0]
        if in_place:
            Lnew, ld_new = L_old, L_old.shape[0]
            touched.append(L_old)
            Lnew[n_old:n, n_old:] = 0.0                               # (the identity rows of the padding these rows were)
        else:
            # a factor that has outgrown its buffer once will be appended to again: room for more rows -- max(256, n / 32), at most
            # 1024 (+2 % of an N = 50k factor, not +6 %); evaluations at a new theta keep a scratch of their own size (_eval_dim)
            ld_new = _lib.pad128(ld_new + min(max(256, n // 32), 1024))
            Lnew = H.zeros(ld_new, ld_new)
            Lnew[:n_old, :n_old] = L_old[:n_old, :n_old]
            if ld_new > n:
                Lnew[n:, n:].fill_diagonal_(1.0)
        Lnew[n_old:n, :n_old] = B[:n_old, :m].T
        Lnew[n_old:n, n_old:n] = S[:m, :m]
        del B, S
        H.invalidate_factor()                                     # Lnew was filled by copies, not by potrf
        self._np, self._ld, self._L = np_new, ld_new, Lnew
        self._x_dev = H.to_device(self.x_data)
        ymean = self.y_data - mean[:, None]
        self._alpha = H.zeros(np_new, ncol)
        self._alpha[:n] = H.to_device(ymean)
        if ncol <= _lib.MAX_RHS_VEC:
            H.potrs(Lnew, n, self._alpha, ncol)
        else:
            rhs = H.zeros(np_new, _lib.pad128(ncol))
            rhs[:n, :ncol] = self._alpha[:n]
            H.potrs(Lnew, n, rhs, _lib.pad128(ncol))
            self._alpha[:n] = rhs[:n, :ncol]
        self._logdet = H.logdet(Lnew, n)
        quad = H.dot(H.to_device(ymean), self._alpha, n) / ncol
        self._loglik = -0.5 * (quad + self._logdet + n * np.log(2.0 * np.pi))
        self.m, self.V = mean, V
        self._K_host = None
        if not (self._work is not None and self._work.shape[0] == self._eval_dim() and self._alpha_work.shape[0] == np_new):
            self._work = self._work2 = self._alpha_work = None    # (kept when the sizes did not change: no reallocation per append)
        self._refresh_inverse()

    @property
    def K(self):
        """Prior covariance at the current hyperparameters as a host array (fvgp/gp.py:625-627),
        materialised on first access only."""
        if self._K_host is None:
            n = self.point_number
            if self._sharded:
                self._K_host = self._sh.kernel_matrix(self._hps)           # every rank's rows, one all-gather
            elif self._native is not None:
                buf = self._H.empty(n, n + (n & 1))
                self._H.kmat(self._native.kernel_id, self._x_dev, self._x_dev, self._hps, buf)
                self._H.sync()
                self._K_host = buf[:, :n].cpu().numpy().copy()
            else:
                self._K_host = self._host_kernel(self.x_data, self.x_data, self._hps)
        return self._K_host

    @property
    def KVinvY(self):
        if self._sharded:
            return self._sh.alpha[:self.point_number, :self.y_data.shape[1]].cpu().numpy()
        self._H.sync()
        return self._alpha[:self.point_number].cpu().numpy()

    @property
    def Chol_factor(self):
        """tril of the device factor (what np.tril(kv.Chol_factor) is in the reference)."""
        if self._sharded:
            return self._sh.factor_matrix()                                # gathered on request, replicated
        if self._linalg_callables is not None:
            return None                                                    # gp_kv.py:123: never set when linalg_mode is three callables
        self._H.sync()
        n = self.point_number
        return np.tril(self._L[:n, :n].cpu().numpy())

    @property
    def logdet_KV(self):
        return self._logdet

    # ------------------------------------------------------------------------------------------
    # marginal likelihood
    # ------------------------------------------------------------------------------------------
    def log_likelihood(self, hyperparameters=None):
        """fvgp/gp.py:1310-1330.  hyperparameters=None returns the cached value; otherwise a full
        evaluation at the new theta that touches no state (gp_kv.py:574-578)."""
        if hyperparameters is None:
            return self._loglik
        try:
            if self._sharded:
                ll = self._evaluate_sharded(hyperparameters, state=False, keep_factor=False)[0]
            else:
                KV, aw = self._scratch()
                ll, _, _, _ = self._evaluate(hyperparameters, KV, aw, need_alpha=False)
        except Exception as e:
            raise Exception(f"Linear algebra failed for hyperparameters {hyperparameters}: {e}") from e
        return ll

    def neg_log_likelihood(self, hyperparameters=None):
        return -self.log_likelihood(hyperparameters=hyperparameters)

    def _check_sharded_gradient(self):
        """The row-sharded gradient re-evaluates dK/dtheta inside its trace kernel: with a kernel callable there is no device
        formula, and the reference itself refuses a gradient in its distributed mode (gp_marginal_likelihood.py:240)."""
        if self._sharded and self._native is None:
            raise Exception("Can't compute neg_log_likelihood_gradient for a row-sharded GP with a kernel callable "
                            "(use one of the named kernels, or a gradient-free training method such as 'mcmc' or 'global')")

    def neg_log_likelihood_gradient(self, hyperparameters=None, component=0):
        """fvgp/gp.py:1332-1353, gp_marginal_likelihood.py:224-309.
        g_i = 1/2 (tr(KV^-1 dKV_i) - b^T dKV_i b) - dm_i^T b, kernel term dropped where the mean term
        is non-zero (:301-308).  KV^-1 comes from POTRI on the device; dK/dtheta is re-evaluated inside
        the trace kernel, never stored."""
        H, n = self._H, self.point_number
        if self._sharded:
            self._check_sharded_gradient()
            return self._gradient_sharded(hyperparameters, component)
        KV, aw = self._scratch()
        if self._work2 is None:
            self._work2 = H.empty(self._np, self._np)
        hps = self._hps if hyperparameters is None else np.asarray(hyperparameters, dtype=np.float64)
        self._evaluate(hps, KV, aw, use_callables=False)       # the reference's gradient goes around the linalg callables too (gp_marginal_likelihood.py:274)
        ncol = self.y_data.shape[1]
        if self._native is not None:
            g = H.loglik_grad(self._native.kernel_id, self._x_dev, hps, aw, ncol, component, KV, self._work2)
            diag_inv = None
        else:
            # host kernel callable: dK/dtheta_i exists as numbers on the host (user gradient, or the reference's central
            # differences); one N^2 upload per direction, the trace against KV^-1 on the device (fvgp_hip_trace_dot)
            H.potri(KV, n, self._work2)
            H.symmetrize(KV, n)
            b_dev = aw[:n, component]
            g = np.zeros(len(hps))
            for i, dKi in enumerate(self._host_kernel_grads(hps)):
                g[i] = 0.5 * H.trace_dot(KV, H.to_device(dKi), b_dev, n)
            diag_inv = None
        g = np.asarray(g, dtype=np.float64)
        if len(g) < len(hps):
            g = np.concatenate([g, np.zeros(len(hps) - len(g))])
        b = None
        # noise-owned hyperparameters: d/dtheta_i of diag V enters exactly like dK (gp_marginal_likelihood.py:262-267)
        if self._noise_callable is not None:
            dV = self._noise_grad(hps)
            if np.ndim(dV) == 3:
                # matrix-valued noise derivative (gp_marginal_likelihood.py:262-267): 1/2 (tr(KV^-1 dV_i) - b^T dV_i b),
                # KV^-1 (lower, on the device by now) mirrored once
                H.symmetrize(KV, n)
                bd = aw[:n, component]
                for i in range(len(hps)):
                    if np.any(dV[i] != 0.0):
                        g[i] += 0.5 * H.trace_dot(KV, H.to_device(dV[i]), bd, n)
            elif np.any(dV != 0.0):
                H.sync()
                b = aw[:n, component].cpu().numpy()
                diag_inv = KV[:n, :n].diagonal().cpu().numpy() if diag_inv is None else diag_inv
                g = g + 0.5 * (dV @ (diag_inv - b * b))
        # mean-owned hyperparameters (:281,301-308)
        if self._mean_callable is not None:
            dm = self._mean_grad(hps)
            if b is None:
                H.sync()
                b = aw[:n, component].cpu().numpy()
            gm = -(dm @ b)
            g = np.where(gm == 0.0, g, 0.0) + gm
        return g

    def _gradient_sharded(self, hyperparameters, component):
        """The same gradient on the row-sharded factor (dist.ShardedGP.gradient); mean-owned hyperparameters as in the
        single-GPU path (:281,301-308).  Noise-function hyperparameters (diagonal noise models, :262-267) take diag(KV^-1)
        from the ranks' rows of inv(L): column sums of squares, summed over the ranks -- no N x N buffer anywhere."""
        n = self.point_number
        if hyperparameters is None:
            hps, sh = self._hps, self._sh
        else:
            hps = np.asarray(hyperparameters, dtype=np.float64)
            sh = self._evaluate_sharded(hps, state=False)[4]
        dV = self._noise_grad(hps) if self._noise_callable is not None else None
        if dV is not None and np.ndim(dV) != 2:
            raise NotImplementedError("the row-sharded mode takes a diagonal noise model")
        if dV is not None and np.any(dV != 0.0):
            g, diag_inv = sh.gradient(component, want_diag=True)
        else:
            g, diag_inv = sh.gradient(component), None
        g = np.asarray(g, dtype=np.float64)
        if len(g) < len(hps):
            g = np.concatenate([g, np.zeros(len(hps) - len(g))])
        if diag_inv is not None:
            b = sh.alpha[:n, component].cpu().numpy()
            g = g + 0.5 * (dV @ (diag_inv - b * b))
        if self._mean_callable is not None:
            b = sh.alpha[:n, component].cpu().numpy()
            gm = -(self._mean_grad(hps) @ b)
            g = np.where(gm == 0.0, g, 0.0) + gm
        return g

    def _host_kernel_grads(self, hps):
        """dK/dtheta_i, i = 0 .. H-1, one (N, N) host array at a time, as GPprior selects it (gp_prior.py:65-74): the user's
        kernel_function_grad -- all directions in one call, or called per direction under ram_economy (:236-240) -- else
        central differences of the kernel itself with the reference's step, eps = 1e-8 (:438-447)."""
        x = self.x_data
        if self._kernel_grad_callable is not None:
            if self.ram_economy:
                for i in range(len(hps)):
                    yield np.ascontiguousarray(self._kernel_grad_callable(x, x, hps, i), dtype=np.float64)
            else:
                dK = self._kernel_grad_callable(x, x, hps)
                for i in range(len(hps)):
                    yield np.ascontiguousarray(dK[i], dtype=np.float64)
            return
        eps = 1e-8
        for i in range(len(hps)):
            hp, hm = np.array(hps, dtype=np.float64), np.array(hps, dtype=np.float64)
            hp[i] += eps
            hm[i] -= eps
            yield (self._host_kernel(x, x, hp) - self._host_kernel(x, x, hm)) / (2.0 * eps)

    def _central_fd(self, f, hps):
        """(H, N) central difference with step 1e-6 -- gp_likelihood.py:123-133, gp_prior.py:460-469."""
        out = []
        for i in range(len(hps)):
            tp, tm = np.array(hps, dtype=np.float64), np.array(hps, dtype=np.float64)
            tp[i] += 1e-6
            tm[i] -= 1e-6
            out.append((f(self.x_data, tp) - f(self.x_data, tm)) / 2e-6)
        return np.array(out)              # (H, N), or (H, N, N) for a matrix-valued noise model

    def _noise_grad(self, hps):
        if self._noise_grad_callable is not None:
            return np.asarray(self._noise_grad_callable(self.x_data, hps), dtype=np.float64)
        return self._central_fd(self._noise, hps)

    def _mean_grad(self, hps):
        if self._mean_grad_callable is not None:
            return np.asarray(self._mean_grad_callable(self.x_data, hps), dtype=np.float64)
        return self._central_fd(self._mean, hps)

    # ------------------------------------------------------------------------------------------
    # posterior
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def cartesian_product(x, y):
        """Task-major product of points and task indices (gp_posterior.py:585-604), vectorised."""
        assert isinstance(y, np.ndarray) and np.ndim(y) == 1, "x_out must be a 1-d np.ndarray for cartesian product"
        x = np.asarray(x, dtype=np.float64)
        return np.hstack([np.tile(x, (len(y), 1)), np.repeat(y.astype(np.float64), len(x))[:, None]])

    def _perform_input_checks(self, x_pred, x_out):
        assert isinstance(x_pred, np.ndarray), "wrong format in x_pred"
        assert np.ndim(x_pred) == 2, "wrong dim in x_pred, has to be 2-d"
        assert isinstance(x_out, np.ndarray) or x_out is None, "wrong format in x_out"
        if isinstance(x_out, np.ndarray):
            assert np.ndim(x_out) == 1, "wrong dim in x_out, has to be 1-d"

    def _posterior_device(self, x_pred, hps, L, alpha, want_cov):
        """k(x_data, x_pred) assembly, mean = k^T alpha, S = kk - k^T KV^-1 k on the device."""
        if self._sharded:
            sh = self._sh if L is None else L
            return sh.posterior(x_pred, want_cov=want_cov)
        H, n = self._H, self.point_number
        P = len(x_pred)
        Pp = _lib.pad128(P)
        ncol = self.y_data.shape[1]
        xp = H.to_device(x_pred)
        if self._linalg_callables is not None:
            # kv.solve through the user's f_solve on the kept factor object (gp_kv.py:697-698; gp_posterior.py:120-136,158)
            obj = self._custom_obj if L is self._L else self._custom_obj_work
            k = (np.asarray(self._native(self.x_data, x_pred, hps)) if self._native is not None
                 else self._host_kernel(self.x_data, x_pred, hps))
            H.sync()
            mean_h = k.T @ alpha[:n].cpu().numpy()
            if not want_cov:
                return mean_h, None
            kk = (np.asarray(self._native(x_pred, x_pred, hps)) if self._native is not None
                  else self._host_kernel(x_pred, x_pred, hps))
            return mean_h, kk - k.T @ np.asarray(self._linalg_callables[1](obj, k), dtype=np.float64).reshape(k.shape)
        if self._native is not None and want_cov and P > self._posterior_chunk:
            return self._posterior_chunked(x_pred, hps, L, alpha)
        mean = H.empty(P, ncol)
        kx = H.empty(self._np, Pp)
        if self._native is not None:
            S = H.empty(Pp, Pp) if want_cov else None
            H.posterior(self._native.kernel_id, self._x_dev, hps, L, alpha, ncol, xp, kx, mean, None, S)
            H.sync()
            return mean.cpu().numpy(), (None if S is None else H.to_host(S[:P, :P]))
        # slow path: host cross-covariances, device solves
        kx.zero_()
        kx[:n, :P] = H.to_device(self._host_kernel(self.x_data, x_pred, hps))
        aw = H.zeros(self._np, _lib.pad128(ncol))
        aw[:n, :ncol] = alpha[:n]
        mw = H.empty(Pp, _lib.pad128(ncol))
        H.gemm(1, 1, 0, Pp, _lib.pad128(ncol), self._np, 1.0, kx, aw, 0.0, mw)           # k^T KVinvY
        mean_h = mw[:P, :ncol].cpu().numpy()
        if not want_cov:
            return mean_h, None
        H.trsm_lower(L, n, kx, Pp)
        S = H.zeros(Pp, Pp)
        S[:P, :P] = H.to_device(self._host_kernel(x_pred, x_pred, hps))
        H.gemm(1, 1, 0, Pp, Pp, self._np, -1.0, kx, kx, 1.0, S)                            # kk - v^T v, v = L^-1 k
        H.sync()
        return mean_h, S[:P, :P].cpu().numpy()

    def _posterior_chunked(self, x_pred, hps, L, alpha):
        """Posterior mean and covariance at MANY prediction points with bounded device memory (gp_posterior.py:120-136,229-288 form
        k (N x P), L^-1 k and the P x P result in one piece each).  The points go through the device in chunks of
        `posterior_chunk` (4096): fvgp_hip_posterior per chunk gives its mean, its diagonal block of S and leaves
        V_i^T = (L^-1 k_i)^T in the chunk's scratch; the off-diagonal blocks are S_ij = k(x_i, x_j) - V_i^T V_j, one MFMA product
        each.  S is assembled on the device while P x P doubles fit a third of the budget (one copy to the host at the end), else
        on the host block by block (the device then never holds more than a chunk x chunk piece of it).  The V_i stay resident
        while N x P doubles fit `posterior_scratch_bytes` (default: a third of the free device memory); beyond that the chunks
        are walked in groups and a group's V is recomputed for every earlier group it meets."""
        H, n, ncol = self._H, self.point_number, self.y_data.shape[1]
        torch = H.torch
        P, C, kid = len(x_pred), int(self._posterior_chunk), self._native.kernel_id
        assert C % 128 == 0 and C >= 128, "posterior_chunk must be a multiple of 128"
        spans = [(s0, min(s0 + C, P)) for s0 in range(0, P, C)]
        Cp = _lib.pad128(C)
        Pp = _lib.pad128(P)
        budget = self._posterior_scratch_bytes
        if budget is None:
            budget = torch.cuda.mem_get_info(H.device)[0] // 3
        on_device = 8 * Pp * Pp <= budget // 3
        vbudget = budget - (8 * Pp * Pp if on_device else 2 * 8 * Cp * Cp)
        fit = max(2, int(vbudget // (self._np * Cp * 8)))                     # chunk scratches that may be resident at once
        gsz = len(spans) if len(spans) <= fit else max(1, fit // 2)           # chunks per group (two groups resident when walking pairs)
        groups = [list(range(g0, min(g0 + gsz, len(spans)))) for g0 in range(0, len(spans), gsz)]
        self._posterior_groups = len(groups)                                  # (diagnostic: 1 = nothing was recomputed)
        mean_h = np.empty((P, ncol))
        S_dev = H.empty(Pp, Pp) if on_device else None
        S_h = None if on_device else np.empty((P, P))
        xp_dev = [H.to_device(x_pred[a:b]) for a, b in spans]
        Sblk = kk = None
        if not on_device:
            Sblk, kk = H.empty(Cp, Cp), H.empty(Cp, Cp)
        mean_d = H.empty(C, ncol)
        var_d = H.empty(C)

        def new_buf():
            return H.empty(self._np * Cp)                                     # flat: handed over as (np x Pp_i), read back as (Pp_i x np)

        def vt(i, buf):
            """the chunk's V_i^T as the call left it: pad128(P_i) x np, leading dimension np"""
            pi = _lib.pad128(spans[i][1] - spans[i][0])
            return buf[:pi * self._np].view(pi, self._np)

        def sweep(i, buf, first):
            """buf <- V_i^T; first: also the chunk's mean and diagonal block"""
            a, b = spans[i]
            pi = _lib.pad128(b - a)
            kx = buf[:self._np * pi].view(self._np, pi)                       # the ABI's scratch: padded N x padded P_i
            if first:
                out = S_dev[a:, a:] if on_device else Sblk
                H.posterior(kid, self._x_dev, hps, L, alpha, ncol, xp_dev[i], kx, mean_d, None, out)
                mean_h[a:b] = mean_d[:b - a].cpu().numpy()
                if not on_device:
                    S_h[a:b, a:b] = H.to_host(Sblk[:b - a, :b - a])
            else:
                H.posterior(kid, self._x_dev, hps, L, alpha, ncol, xp_dev[i], kx, None, var_d, None)

        def cross(i, bi, j, bj):
            """S[i-rows, j-cols] = k(x_i, x_j) - V_i^T V_j  (i > j); the upper half is mirrored at the end"""
            (a, b), (c, e) = spans[i], spans[j]
            pi, pj = _lib.pad128(b - a), _lib.pad128(e - c)
            out = S_dev[a:, c:] if on_device else kk
            H.kmat(kid, xp_dev[i], xp_dev[j], hps, out, pad=_lib.PAD_ZERO)
            H.gemm(0, 0, 0, pi, pj, self._np, -1.0, vt(i, bi), vt(j, bj), 1.0, out)
            if not on_device:
                blk = H.to_host(kk[:b - a, :e - c])
                S_h[a:b, c:e] = blk
                S_h[c:e, a:b] = blk.T

        for ga, grp_a in enumerate(groups):
            bufs_a = {i: new_buf() for i in grp_a}
            for i in grp_a:
                sweep(i, bufs_a[i], True)
                for j in grp_a:
                    if j < i:
                        cross(i, bufs_a[i], j, bufs_a[j])
            for grp_b in groups[ga + 1:]:
                for i in grp_b:                                              # a later group's chunks, one scratch at a time
                    bi = new_buf()
                    sweep(i, bi, False)
                    for j in grp_a:
                        cross(i, bi, j, bufs_a[j])
                    del bi
            del bufs_a
        if on_device:
            H.symmetrize(S_dev, P)
            H.sync()
            return mean_h, H.to_host(S_dev[:P, :P])
        H.sync()
        return mean_h, S_h

    def _variance_from_inverse(self, x_pred):
        H, n = self._H, self.point_number
        P = len(x_pred)
        Pp = _lib.pad128(P)
        kx = H.empty(self._np, Pp)
        H.kmat(self._native.kernel_id, self._x_dev, H.to_device(x_pred), self._hps, kx, pad=_lib.PAD_ZERO)
        Wk = H.empty(self._np, Pp)
        H.gemm(0, 1, 0, self._np, Pp, self._np, 1.0, self._KVinv, kx, 0.0, Wk)        # KVinv @ k on MFMA
        q = H.empty(P)
        H.coldot(kx, Wk, n, P, q)                                                     # k_p^T KVinv k_p per prediction point
        H.sync()
        return self._hps[0] - q.cpu().numpy()

    def posterior_mean(self, x_pred, hyperparameters=None, x_out=None):
        """fvgp/gp.py:1376-1431, gp_posterior.py:139-182."""
        if self._sharded:
            L, alpha, hps = None, None, self._hps
            if hyperparameters is not None:
                hps = np.asarray(hyperparameters, dtype=np.float64)
                L = self._evaluate_sharded(hps, state=False)[4]            # the scratch twin, factored at hps
        else:
            L, alpha, hps = self._L, self._alpha, self._hps
            if hyperparameters is not None:
                hps = np.asarray(hyperparameters, dtype=np.float64)
                L, alpha = self._scratch()
                self._evaluate(hps, L, alpha)
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = self.cartesian_product(x_pred, x_out)
        assert x_pred.shape[1] == self.index_set_dim, "wrong number of columns in x_pred"
        A, _ = self._posterior_device(x_pred, hps, L, alpha, want_cov=False)
        posterior_mean = self._mean(x_pred, hps)[:, None] + A
        ncol = self.y_data.shape[1]
        if isinstance(x_out, np.ndarray):
            posterior_mean_re = posterior_mean.reshape(len(x_orig), len(x_out), order='F')
        else:
            posterior_mean_re = posterior_mean
        if ncol == 1 and not isinstance(x_out, np.ndarray):
            return {"x": x_orig, "m(x)": np.squeeze(posterior_mean_re), "m(x)_flat": np.squeeze(posterior_mean),
                    "x_pred": x_pred}
        if ncol == 1:
            return {"x": x_orig, "m(x)": posterior_mean_re, "m(x)_flat": np.squeeze(posterior_mean), "x_pred": x_pred}
        return {"x": x_orig, "m(x)": posterior_mean_re, "m(x)_flat": posterior_mean, "x_pred": x_pred}

    def posterior_covariance(self, x_pred, x_out=None, variance_only=False, add_noise=False):
        """fvgp/gp.py:1433-1480, gp_posterior.py:229-288 (Chol mode: S is always formed, :246)."""
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = self.cartesian_product(x_pred, x_out)
        assert x_pred.shape[1] == self.index_set_dim, "wrong number of columns in x_pred"
        if self._KVinv is not None and variance_only and self.y_data.shape[1] == 1 and self._native is not None:
            # gp_posterior.py:238-244: v = diag(kk) - einsum('ij,jk,ki->i', k^T, KVinv, k), S never formed
            S = None
            v = self._variance_from_inverse(x_pred)
        else:
            _, S = self._posterior_device(x_pred, self._hps, None if self._sharded else self._L,
                                          None if self._sharded else self._alpha, want_cov=True)
            v = np.array(np.diag(S))
        if np.any(v < -0.0001):
            warnings.warn("Negative variances encountered. That normally means that the model is unstable. "
                          "Rethink the kernel definition, add more noise to the data, "
                          "or double check the hyperparameter optimization bounds. This will not "
                          "terminate the algorithm, but expect anomalies.")
        if np.any(v < 0.0):
            v[v < 0.0] = 0.0
            if not variance_only:
                np.fill_diagonal(S, v)
        if add_noise:
            noise = self._noise(x_pred, self._hps)          # gp_posterior.py:554-569
            if np.ndim(noise) == 2:
                v = v + np.diag(noise)
                if S is not None:
                    S = S + noise
            else:
                v = v + noise
                if S is not None:
                    S = S + np.diag(noise)
        if isinstance(x_out, np.ndarray):
            v_re = v.reshape(len(x_orig), len(x_out), order='F')
            S_re = None if S is None else \
                S.reshape(len(x_orig), len(x_out), len(x_orig), len(x_out), order='F').transpose(0, 2, 1, 3)
        else:
            v_re, S_re = v, S
            if self.y_data.shape[1] > 1:
                v = np.tile(v[:, None], (1, self.y_data.shape[1]))
                v_re = np.tile(v_re[:, None], (1, self.y_data.shape[1]))
        return {"x": x_orig, "x_pred": x_pred, "v(x)": v_re, "S": S_re, "S_flat": S, "v_flat": v}

    # ------------------------------------------------------------------------------------------
    # derivatives built on the path -- finite differences of the same device evaluations, with the
    # reference's steps (so they agree with it to its own noise level, not beyond)
    # ------------------------------------------------------------------------------------------
    def neg_log_likelihood_hessian(self, hyperparameters=None):
        """fvgp/gp.py, gp_marginal_likelihood.py:312-336: forward difference (1e-6) of the exact gradient."""
        hps = self._hps if hyperparameters is None else np.asarray(hyperparameters, dtype=np.float64)
        nh = len(hps)
        d2 = np.zeros((nh, nh))
        epsilon = 1e-6
        g0 = self.neg_log_likelihood_gradient(hyperparameters=hps)
        for i in range(nh):
            t = np.array(hps)
            t[i] = t[i] + epsilon
            d2[i, i:] = ((self.neg_log_likelihood_gradient(hyperparameters=t) - g0) / epsilon)[i:]
        return d2 + d2.T - np.diag(np.diag(d2))

    def test_log_likelihood_gradient(self, hyperparameters, epsilon=1e-6):
        """gp_marginal_likelihood.py:338-364: (forward-difference gradient, analytical gradient) of the log-likelihood."""
        thps = np.array(hyperparameters, dtype=np.float64)
        grad = np.empty(len(thps))
        base = self.log_likelihood(hyperparameters=thps)
        for i in range(len(thps)):
            aux = np.array(thps)
            aux[i] = aux[i] + epsilon
            grad[i] = (self.log_likelihood(hyperparameters=aux) - base) / epsilon
        return grad, -self.neg_log_likelihood_gradient(hyperparameters=thps)

    def _cross_dev(self, x_b, hps):
        """k(x_data, x_b) on the device, (padded N) x (padded P), zero padding."""
        H, n = self._H, self.point_number
        P = len(x_b)
        kx = H.empty(self._np, _lib.pad128(P))
        if self._native is not None:
            H.kmat(self._native.kernel_id, self._x_dev, H.to_device(x_b), hps, kx, pad=_lib.PAD_ZERO)
        else:
            kx.zero_()
            kx[:n, :P] = H.to_device(self._host_kernel(self.x_data, x_b, hps))
        return kx

    def _kk_host(self, x_b, hps):
        """k(x_b, x_b) as a host array (P x P)."""
        if self._native is None:
            return self._host_kernel(x_b, x_b, hps)
        H = self._H
        P = len(x_b)
        xb = H.to_device(x_b)
        buf = H.empty(P, P + (P & 1))
        H.kmat(self._native.kernel_id, xb, xb, hps, buf)
        H.sync()
        return buf[:, :P].cpu().numpy()

    def posterior_mean_grad(self, x_pred, hyperparameters=None, x_out=None, direction=None, component=0):
        """fvgp/gp.py, gp_posterior.py:184-226: dm/dx = d(prior mean)/dx (step 1e-6) + dk/dx^T KVinvY with the
        kernel derivative taken by a forward difference of step 1e-8 (gp_prior.py:402-409).  By linearity
        dk/dx^T KVinvY is the same difference of two device evaluations of k^T KVinvY."""
        if self._sharded:
            raise NotImplementedError("the finite-difference posterior derivatives run on the single-GPU path only")
        L, alpha, hps = self._L, self._alpha, self._hps
        if hyperparameters is not None:
            hps = np.asarray(hyperparameters, dtype=np.float64)
            L, alpha = self._scratch()
            self._evaluate(hps, L, alpha)
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = self.cartesian_product(x_pred, x_out)
        f = self._mean(x_pred, hps)
        eps = 1e-6
        A0 = self._posterior_device(x_pred, hps, L, alpha, want_cov=False)[0][:, component]

        def one(dd):
            x1 = np.array(x_pred)
            x1[:, dd] = x1[:, dd] + eps
            mean_der = (self._mean(x1, hps) - f) / eps
            xk = np.array(x_pred)
            xk[:, dd] += 1e-8
            A1 = self._posterior_device(xk, hps, L, alpha, want_cov=False)[0][:, component]
            return mean_der + (A1 - A0) / 1e-8

        if direction is not None:
            g = one(direction)
            if isinstance(x_out, np.ndarray):
                g = g.reshape(len(x_orig), len(x_out), order='F')
        else:
            g = np.zeros((len(x_pred), x_orig.shape[1]))
            for dd in range(len(x_orig[0])):
                g[:, dd] = one(dd)
            direction = "ALL"
            if isinstance(x_out, np.ndarray):
                g = g.reshape(len(x_orig), len(x_orig[0]), len(x_out), order='F')
        return {"x": x_orig, "direction": direction, "dm/dx": g}

    def posterior_covariance_grad(self, x_pred, x_out=None, direction=None):
        """fvgp/gp.py, gp_posterior.py:290-331: dS/dx = dkk/dx (step 1e-6) - 2 dk/dx^T KV^-1 k (kernel step 1e-8).
        KV^-1 k is one device solve; the two cross products run on the MFMA GEMM."""
        if self._sharded:
            raise NotImplementedError("the finite-difference posterior derivatives run on the single-GPU path only")
        H, n = self._H, self.point_number
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = self.cartesian_product(x_pred, x_out)
        hps = self._hps
        P = len(x_pred)
        Pp = _lib.pad128(P)
        k0 = self._cross_dev(x_pred, hps)
        if self._linalg_callables is not None:
            # kv.solve through the user's f_solve on the kept factor object (gp_kv.py:697-698), as the posterior does: the
            # device buffer holds K + V here, not a Cholesky factor
            H.sync()
            kh = k0[:n, :P].cpu().numpy()
            Wh = np.zeros((self._np, Pp))
            Wh[:n, :P] = np.asarray(self._linalg_callables[1](self._custom_obj, kh), dtype=np.float64).reshape(kh.shape)
            W = H.to_device(Wh)
        else:
            W = k0.clone()
            H.potrs(self._L, n, W, Pp)                               # KV^-1 k
        kk0 = self._kk_host(x_pred, hps)
        C = H.empty(Pp, Pp)
        eps = 1e-6

        def dS(dd):
            xk = np.array(x_pred)
            xk[:, dd] += 1e-8
            k1 = self._cross_dev(xk, hps)
            H.gemm(1, 1, 0, Pp, Pp, self._np, 1.0 / 1e-8, k1, W, 0.0, C)          # (k1 - k0)^T W / 1e-8
            H.gemm(1, 1, 0, Pp, Pp, self._np, -1.0 / 1e-8, k0, W, 1.0, C)
            H.sync()
            x1 = np.array(x_pred)
            x1[:, dd] = x1[:, dd] + eps
            kk_g = (self._kk_host(x1, hps) - kk0) / eps
            return kk_g - 2.0 * C[:P, :P].cpu().numpy()

        if direction is not None:
            dSdx = dS(direction)
            a = np.diag(dSdx)
            if isinstance(x_out, np.ndarray):
                a = a.reshape(len(x_orig), len(x_out), order='F')
                dSdx = dSdx.reshape(len(x_orig), len(x_orig), len(x_out), len(x_out), order='F')
            return {"x": x_orig, "dv/dx": a, "dS/dx": dSdx}
        grad_v = np.zeros((len(x_pred), len(x_orig[0])))
        for dd in range(len(x_orig[0])):
            grad_v[:, dd] = np.diag(dS(dd))
        if isinstance(x_out, np.ndarray):
            grad_v = grad_v.reshape(len(x_orig), len(x_orig[0]), len(x_out), order='F')
        return {"x": x_orig, "dv/dx": grad_v}

    # ------------------------------------------------------------------------------------------
    # training: the callers of the path -- device-resident objective, host optimiser
    # ------------------------------------------------------------------------------------------
    def train(self, hyperparameter_bounds=None, objective_function=None, objective_function_gradient=None,
              objective_function_hessian=None, init_hyperparameters=None, method="mcmc", pop_size=20, tolerance=0.0001,
              max_iter=10000, mcmc_prior=None, mcmc_prop_distrs="normal", mcmc_args=None, bo_args=None,
              local_optimizer="L-BFGS-B", global_optimizer="genetic", constraints=(), dask_client=None, info=False,
              asynchronous=False, accept_only_if_improved=True, seed=None):
        """fvgp/gp.py:781-1141 for the methods that run without Dask/HGDL: 'mcmc' (default), 'adam',
        'global' (differential evolution), 'local', or a callable that gets the GP.  Every objective call is one
        device evaluation; x, y never leave HBM.  Returns the optimised hyperparameters and sets them (gp.py:1112).
        Argument checks follow gp.py:1007-1053: default bounds come with a warning, out-of-bounds starting points are
        redrawn uniformly with a warning, 'mcmc' ignores a user objective, a user objective for 'local' needs its
        gradient.  'mcmc' draws from numpy's legacy global stream like the reference (seed: a RandomState(seed)
        instead); 'local' / 'adam' / callable results that lower the log marginal likelihood are rejected unless
        accept_only_if_improved=False (gp.py:1086-1168)."""
        from . import gp_training
        if asynchronous:
            if dask_client is None:
                raise Exception("Please provide a dask_client for asynchronous training")
            raise NotImplementedError("asynchronous (Dask actor) training is outside this engine's scope")
        if hyperparameter_bounds is None:
            hyperparameter_bounds = self._default_bounds()
            warnings.warn("Default hyperparameter_bounds initialized because none were provided. "
                          "This will fail for custom kernel, mean, or noise functions")
        hyperparameter_bounds = np.asarray(hyperparameter_bounds, dtype=np.float64)
        if self._sharded and seed is None:
            # every rank walks the same optimiser trajectory (each objective call is a collective): one seed for all
            import torch
            import torch.distributed as dist
            pg = self.args.get("process_group")
            t = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
            if dist.is_initialized():
                dev = "cuda" if dist.get_backend(None if pg is True else pg) == "nccl" else "cpu"
                t = t.to(dev)
                dist.broadcast(t, src=0, group=None if pg is True else pg)
            seed = int(t.item())

        def redraw():                                                   # gp.py:1026-1036
            rng = np.random if seed is None else np.random.RandomState(seed)
            return rng.uniform(low=hyperparameter_bounds[:, 0], high=hyperparameter_bounds[:, 1],
                               size=len(hyperparameter_bounds))
        if init_hyperparameters is None:
            init_hyperparameters = self._hps.copy()
            if len(init_hyperparameters) == len(hyperparameter_bounds) and \
                    not gp_training._in_bounds(init_hyperparameters, hyperparameter_bounds):
                init_hyperparameters = redraw()
        else:
            init_hyperparameters = np.asarray(init_hyperparameters, dtype=np.float64)
            if len(init_hyperparameters) == len(hyperparameter_bounds) and \
                    not gp_training._in_bounds(init_hyperparameters, hyperparameter_bounds):
                warnings.warn("Your init_hyperparameters are out of bounds. They will be over-written")
                init_hyperparameters = redraw()
        user_objective = objective_function is not None
        if method == "mcmc" and user_objective:
            warnings.warn("MCMC always optimizes the log marginal likelihood; "
                          "the user-defined objective_function is ignored.")
            objective_function = None
        if not user_objective and method in ("local", "adam", "hgdl"):
            self._check_sharded_gradient()                                 # up front, on every rank, not in the middle of the optimisation
        if user_objective and objective_function_gradient is None and method in ("local", "hgdl"):
            raise Exception("A gradient (and Hessian) of the objective function must be provided "
                            "for method='local' or method='hgdl'.")
        guarded = (accept_only_if_improved and not user_objective and
                   (callable(method) or method in ("local", "hgdl", "adam")))
        incumbent = self._hps.copy() if guarded else None
        ll_incumbent = self.log_likelihood() if guarded else None
        hps = gp_training.train(self, hyperparameter_bounds, init_hyperparameters, method=method,
                                pop_size=pop_size, tolerance=tolerance, max_iter=max_iter,
                                local_optimizer=local_optimizer, constraints=constraints, info=info, seed=seed,
                                objective_function=objective_function,
                                objective_function_gradient=objective_function_gradient,
                                objective_function_hessian=objective_function_hessian,
                                mcmc_prior=mcmc_prior, mcmc_args={} if mcmc_args is None else mcmc_args,
                                mcmc_prop_distrs=mcmc_prop_distrs)
        self.set_hyperparameters(np.asarray(hps, dtype=np.float64))
        if guarded and not self.log_likelihood() >= ll_incumbent:          # exact mode: strict comparison (gp.py:1158-1160)
            warnings.warn(f"Training with method=`{method}` returned hyperparameters with a lower log marginal likelihood "
                          f"({self.log_likelihood()} vs. {ll_incumbent}); they were rejected and the previous hyperparameters "
                          "kept. Pass `accept_only_if_improved=False` to accept them anyway.")
            self.set_hyperparameters(incumbent)
        return self._hps

    def _default_bounds(self):
        """gp.py:752-774: signal variance from the data variance, length scales from the data ranges."""
        if (self._native is None or self._native.isotropic or self._mean_callable is not None
                or self._noise_callable is not None or len(self._hps) != self.index_set_dim + 1):
            raise Exception("Please provide custom hyperparameter_bounds when kernel, mean or noise"
                            " functions are customized")
        b = np.zeros((self.index_set_dim + 1, 2))
        b[0] = np.array([np.var(self.y_data) / 100., np.var(self.y_data) * 10.])
        for i in range(self.index_set_dim):
            range_xi = np.max(self.x_data[:, i]) - np.min(self.x_data[:, i])
            b[i + 1] = np.array([range_xi / 100., range_xi * 10.])
        return b

    # ------------------------------------------------------------------------------------------
    # pickling (fvgp/gp.py:2253-2266, gp_kv.py:718-765): host copies of the state, factor included
    # ------------------------------------------------------------------------------------------
    def __getstate__(self):
        if self._sharded:
            # one slice of the factor per rank: the pickle carries the data and the hyperparameters, the factor is rebuilt by
            # the (collective) re-evaluation at first use after unpickling (fvgp/gp_data.py:147, gp_prior.py:494 drop the Dask
            # client the same way); a ProcessGroup object does not pickle -- the default group takes its place
            st = {k: v for k, v in self.__dict__.items() if k not in ("_H", "_sh", "_sh_work", "_K_host", "args")}
            st["args"] = {k: (True if k == "process_group" else v) for k, v in self.args.items() if k != "shard_ops"}
            st["_sharded_pickle"] = True
            return st
        self._H.sync()
        st = {k: v for k, v in self.__dict__.items()
              if k not in ("_H", "_x_dev", "_L", "_alpha", "_work", "_work2", "_alpha_work", "_KVinv", "_custom_obj", "_custom_obj_work")}
        n = self.point_number
        st["_L_host"] = np.tril(self._L[:n, :n].cpu().numpy())
        st["_alpha_host"] = self._alpha[:n].cpu().numpy()
        return st

    def __setstate__(self, st):
        if st.pop("_sharded_pickle", False):
            self.__dict__.update(st)
            self._H = self._sh = self._sh_work = self._K_host = None
            self.set_hyperparameters(self._hps)                       # collective: every rank unpickles
            return
        L_host, a_host = st.pop("_L_host"), st.pop("_alpha_host")
        self.__dict__.update(st)
        for k, v in (("_posterior_chunk", 4096), ("_posterior_scratch_bytes", None), ("_posterior_groups", 0)):
            self.__dict__.setdefault(k, v)                            # (objects pickled before the chunked posterior)
        self._H = default_handle()
        if self.__dict__.get("_linalg_callables") is not None:       # the user's factor object does not travel: rebuild the state
            self._set_data(self.x_data, self.y_data, self.noise_variances)
            self.set_hyperparameters(self._hps)
            return
        H, n = self._H, self.point_number
        self._x_dev = H.to_device(self.x_data)
        self._ld = self.__dict__.get("_ld", self._np)                # (objects pickled before the buffers had their extra block row)
        self._L = H.zeros(self._ld, self._ld)
        self._L[:n, :n] = H.to_device(L_host)
        if self._ld > n:
            self._L[n:, n:] = H.to_device(np.eye(self._ld - n))
        self._alpha = H.zeros(self._np, self.y_data.shape[1])
        self._alpha[:n] = H.to_device(a_host)
        H.invalidate_factor()                                     # uploaded factor: no cached block inverses belong to it
        self._work = self._work2 = self._alpha_work = None
        self._KVinv = None
        self._refresh_inverse()
