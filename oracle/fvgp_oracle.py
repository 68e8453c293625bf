"""CPU oracle for the exact-GP hot path of lbl-camera/fvGP  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of the reference's dense path
(covariance assembly -> K+V -> Cholesky -> solve / log-det -> log marginal
likelihood, its gradient, and the posterior).  It exists so the HIP kernels can
be checked on a machine that does not have the reference checked out.

Rules ("Oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import it; nothing under fvgp_amd/ does;
  * every function cites the reference file:line it restates (paths relative to
    the reference root);
  * parity is PINNED: oracle/make_golden.py imported the reference itself in the
    build container and froze its outputs under tests/golden/*.npz;
    tests/test_oracle_golden.py holds this module to those vectors.

Third-party arithmetic: the reference's Cholesky / triangular solves are
scipy.linalg.cho_factor / cho_solve (LAPACK dpotrf / dpotrs via OpenBLAS),
fvgp/gp_lin_alg.py:20,245,299 -- pyproject pins scipy>=1.13, numpy>=2.1; this
image has scipy 1.15.3 / numpy 2.2.6, and the oracle calls the same routines.
"""
import warnings

import numpy as np
from scipy.linalg import cho_factor, cho_solve

SQRT3 = np.sqrt(3.0)
SQRT5 = np.sqrt(5.0)


# --------------------------------------------------------------------------
# distance matrices                                     fvgp/kernels.py:440-481
# --------------------------------------------------------------------------
def get_distance_matrix(x1, x2):
    """Isotropic Euclidean distance, fvgp/kernels.py:440-458."""
    acc = np.zeros((len(x1), len(x2)))
    for k in range(x1.shape[1]):
        acc += (x1[:, k].reshape(-1, 1) - x2[:, k]) ** 2
    return np.sqrt(acc)


def get_anisotropic_distance_matrix(x1, x2, lengths):
    """Axis-scaled distance r_ij = sqrt(sum_k ((x1_ik-x2_jk)/l_k)^2), fvgp/kernels.py:461-481."""
    acc = np.zeros((len(x1), len(x2)))
    for k in range(len(x1[0])):
        acc += abs(np.subtract.outer(x1[:, k], x2[:, k]) / lengths[k]) ** 2
    return np.sqrt(acc)


# --------------------------------------------------------------------------
# radial functions                               fvgp/kernels.py:16-33,98-118,166-188
# --------------------------------------------------------------------------
def squared_exponential_kernel(distance, length):
    """exp(-r^2 / (2 l^2)), fvgp/kernels.py:16-33."""
    return np.exp(-(distance ** 2) / (2.0 * (length ** 2)))


def matern_kernel_diff1(distance, length):
    """(1 + sqrt3 r/l) exp(-sqrt3 r/l), fvgp/kernels.py:98-118."""
    return (1.0 + ((SQRT3 * distance) / length)) * np.exp(-(SQRT3 * distance) / length)


def matern_kernel_diff2(distance, length):
    """(1 + sqrt5 r/l + 5 r^2/(3 l^2)) exp(-sqrt5 r/l), fvgp/kernels.py:166-188."""
    return (1.0 + ((SQRT5 * distance) / length) + ((5.0 * distance ** 2) / (3.0 * length ** 2))) \
        * np.exp(-(SQRT5 * distance) / length)


# --------------------------------------------------------------------------
# the named kernels the native path implements
#   theta = [signal variance, l_1 .. l_d]  (ARD)   or   [signal variance, l]  (iso)
# --------------------------------------------------------------------------
def rbf_ard(x1, x2, hps):
    """hps[0] * squared_exponential_kernel(aniso distance, 1)  -- SURVEY 8a row 2 canonical RBF."""
    return hps[0] * squared_exponential_kernel(get_anisotropic_distance_matrix(x1, x2, hps[1:]), 1.0)


def matern32_ard(x1, x2, hps):
    """The reference default kernel, fvgp/gp_prior.py:376-400."""
    acc = np.zeros((len(x1), len(x2)))
    for k in range(len(x1[0])):
        acc += abs(np.subtract.outer(x1[:, k], x2[:, k]) / hps[1 + k]) ** 2
    return hps[0] * matern_kernel_diff1(np.sqrt(acc), 1)


def matern52_ard(x1, x2, hps):
    """fvgp/gp_bo.py:115-126 (_surrogate_kernel)."""
    return hps[0] * matern_kernel_diff2(get_anisotropic_distance_matrix(x1, x2, hps[1:]), 1.0)


def rbf_iso(x1, x2, hps):
    """Notebook form: hps[0] * sq_exp(get_distance_matrix, hps[1]); tests/test_fvgp.py:218-220 style."""
    return hps[0] * squared_exponential_kernel(get_distance_matrix(x1, x2), hps[1])


def matern32_iso(x1, x2, hps):
    return hps[0] * matern_kernel_diff1(get_distance_matrix(x1, x2), hps[1])


def matern52_iso(x1, x2, hps):
    return hps[0] * matern_kernel_diff2(get_distance_matrix(x1, x2), hps[1])


KERNELS = {
    "rbf_ard": rbf_ard, "matern32_ard": matern32_ard, "matern52_ard": matern52_ard,
    "rbf_iso": rbf_iso, "matern32_iso": matern32_iso, "matern52_iso": matern52_iso,
}


# --------------------------------------------------------------------------
# kernel theta-gradients, shape (H, N1, N2)
# --------------------------------------------------------------------------
def matern32_ard_grad(x1, x2, hps):
    """Analytic gradient of the default kernel, fvgp/gp_prior.py:421-436 with
    matern_kernel_diff1_grad (fvgp/kernels.py:121-141): d/dl_i = s*3*D_i^2/l_i^3*exp(-sqrt3 r),
    zero where r == 0."""
    grad = np.zeros((len(hps), len(x1), len(x2)))
    acc = np.zeros((len(x1), len(x2)))
    for k in range(len(x1[0])):
        acc += abs(np.subtract.outer(x1[:, k], x2[:, k]) / hps[1 + k]) ** 2
    r = np.sqrt(acc)
    nz = np.where(r != 0.0)
    for k in range(len(x1[0])):
        drdl = np.zeros(r.shape)
        drdl[nz] = -abs(np.subtract.outer(x1[:, k], x2[:, k]))[nz] ** 2 / (hps[k + 1] ** 3 * r[nz])
        a = SQRT3 * r
        dadl = SQRT3 * drdl
        ea = np.exp(-a)
        grad[k + 1] = hps[0] * (dadl * ea - (1.0 + a) * dadl * ea)
    grad[0] = matern_kernel_diff1(r, 1)
    return grad


def matern52_ard_grad(x1, x2, hps):
    """fvgp/gp_bo.py:167-201 (_surrogate_kernel_grad)."""
    dim = x1.shape[1]
    lengths = np.asarray(hps[1:1 + dim], dtype=float)
    diff = x1[:, None, :] - x2[None, :, :]
    r = np.sqrt(np.sum((diff / lengths[None, None, :]) ** 2, axis=2))
    decay = np.exp(-SQRT5 * r)
    grad = np.zeros((len(hps), x1.shape[0], x2.shape[0]))
    grad[0] = (1.0 + SQRT5 * r + (5.0 / 3.0) * r ** 2) * decay
    common = (5.0 / 3.0) * hps[0] * (1.0 + SQRT5 * r) * decay
    for i in range(dim):
        grad[1 + i] = common * diff[:, :, i] ** 2 / lengths[i] ** 3
    return grad


def rbf_ard_grad(x1, x2, hps):
    """Not in the reference tree: derived, d/dl_i = k * D_i^2 / l_i^3.
    Checked against the reference's own central-FD route (gp_prior.py:438-447) in make_golden.py."""
    dim = x1.shape[1]
    k = rbf_ard(x1, x2, hps)
    grad = np.zeros((len(hps),) + k.shape)
    grad[0] = k / hps[0]
    for i in range(dim):
        grad[1 + i] = k * np.subtract.outer(x1[:, i], x2[:, i]) ** 2 / hps[1 + i] ** 3
    return grad


def fd_kernel_grad(kernel, x1, x2, hps, eps=1e-8):
    """Central finite difference of the reference, fvgp/gp_prior.py:411-419,438-447."""
    grad = np.empty((len(hps), len(x1), len(x2)))
    for i in range(len(hps)):
        hp, hm = np.array(hps, dtype=float), np.array(hps, dtype=float)
        hp[i] += eps
        hm[i] -= eps
        grad[i] = (kernel(x1, x2, hp) - kernel(x1, x2, hm)) / (2.0 * eps)
    return grad


KERNEL_GRADS = {"rbf_ard": rbf_ard_grad, "matern32_ard": matern32_ard_grad, "matern52_ard": matern52_ard_grad}


# --------------------------------------------------------------------------
# K + V, Cholesky, solve, log-det              fvgp/gp_kv.py:639-669, gp_lin_alg.py:237-360
# --------------------------------------------------------------------------
class NonPositiveDefiniteError(np.linalg.LinAlgError):
    """fvgp/gp_lin_alg.py:27-29."""


def addKV(K, V):
    """KV = K + diag(V) (V 1-d) or K + V (V 2-d), fvgp/gp_kv.py:654-667."""
    if np.ndim(V) == 2:
        return K + V
    KV = K.copy()
    np.fill_diagonal(KV, np.diag(K) + V)
    return KV


def calculate_Chol_factor(M):
    """cho_factor(lower=True); strict upper triangle is unspecified. fvgp/gp_lin_alg.py:237-269."""
    try:
        c, _ = cho_factor(M, lower=True)
    except (np.linalg.LinAlgError, RuntimeError) as e:
        raise NonPositiveDefiniteError(
            f"Cholesky factorization failed: the {M.shape[0]}x{M.shape[0]} prior covariance matrix "
            f"is not positive definite. min(diag(M)) = {float(np.min(np.diag(M))):.3e}, "
            f"max|M - M.T| = {float(np.max(np.abs(M - M.T))):.3e}. Original error: {e}") from e
    return c


def calculate_Chol_solve(factor, vec):
    """cho_solve((L, lower), vec); 1-d vec -> (N,1). fvgp/gp_lin_alg.py:289-328."""
    if np.ndim(vec) == 1:
        vec = vec.reshape(len(vec), 1)
    if vec.dtype != factor.dtype:
        vec = vec.astype(factor.dtype, copy=False)
    res = cho_solve((factor, True), vec)
    if np.ndim(res) == 1:
        res = res.reshape(len(res), 1)
    return res


def calculate_Chol_logdet(factor):
    """2 * sum(log|L_ii|), fvgp/gp_lin_alg.py:336-338."""
    return 2.0 * np.sum(np.log(abs(factor.diagonal())))


# --------------------------------------------------------------------------
# multi-task index-set transform            fvgp/fvgp.py:626-660, gp_posterior.py:585-604
# --------------------------------------------------------------------------
def transform_index_set(x_data, y_data, noise_variances=None):
    """(V,Di),(V,No) -> task-major (V*No, Di+1), (V*No,), NaN tasks dropped. fvgp/fvgp.py:626-660."""
    n_out = y_data.shape[1]
    xs, ys, vs = [], [], ([] if noise_variances is not None else None)
    for t in range(n_out):
        for j in range(len(x_data)):
            if np.isnan(y_data[j, t]):
                continue
            xs.append(np.append(x_data[j], float(t)))
            ys.append(y_data[j, t])
            if vs is not None:
                vs.append(noise_variances[j, t])
    return np.asarray(xs), np.asarray(ys), (np.asarray(vs) if vs is not None else None)


def cartesian_product(x, x_out):
    """Task-major product of prediction points and task indices, fvgp/gp_posterior.py:585-604."""
    rows = []
    for t in range(len(x_out)):
        for i in range(len(x)):
            rows.append(np.append(x[i], x_out[t]))
    return np.asarray(rows)


# --------------------------------------------------------------------------
# the GP itself: the reference's call chain restated in one class
# --------------------------------------------------------------------------
class OracleGP:
    """Restates GP.__init__ -> GPprior/GPlikelihood/GPkv state build (fvgp/gp.py:483-568,
    gp_kv.py:404-423) and the four hot methods.  kernel = name in KERNELS (or a callable)."""

    def __init__(self, x_data, y_data, init_hyperparameters, noise_variances=None,
                 kernel="matern32_ard", kernel_grad=None, x_out=None):
        self.x_data = np.asarray(x_data, dtype=float)
        y = np.asarray(y_data, dtype=float)
        if y.ndim == 1:                                   # fvgp/gp_data.py:24
            y = y.reshape(len(y), 1)
        self.y_data = y
        self.noise_variances = noise_variances
        self.kernel = KERNELS[kernel] if isinstance(kernel, str) else kernel
        if kernel_grad is not None:
            self.kernel_grad = kernel_grad
        elif isinstance(kernel, str) and kernel in KERNEL_GRADS:
            self.kernel_grad = KERNEL_GRADS[kernel]
        else:
            self.kernel_grad = lambda a, b, h: fd_kernel_grad(self.kernel, a, b, h)
        self.x_out = x_out
        self.set_hyperparameters(init_hyperparameters)

    # --- pieces -----------------------------------------------------------------
    def mean(self, x):
        """Default prior mean = mean over ALL y entries, fvgp/gp_prior.py:449-458."""
        m = np.zeros(len(x))
        m[:] = np.mean(self.y_data)
        return m

    def noise(self, x):
        """fvgp/gp_likelihood.py:102-110."""
        if self.noise_variances is None:
            return np.ones(len(x)) * (np.mean(abs(self.y_data)) / 100.0) ** 2
        if len(x) == len(self.noise_variances):
            return self.noise_variances
        return np.zeros(len(x)) + np.mean(self.noise_variances)

    def set_hyperparameters(self, hps):
        """fvgp/gp.py:672-687 -> prior / likelihood / kv refresh (gp_kv.py:404-423)."""
        self.hyperparameters = np.asarray(hps, dtype=float)
        self.K = self.kernel(self.x_data, self.x_data, self.hyperparameters)
        self.V = self.noise(self.x_data)
        self.m = self.mean(self.x_data)
        KV = addKV(self.K, self.V)
        self.Chol_factor = calculate_Chol_factor(KV)
        y_mean = self.y_data - self.m[:, None]
        self.KVinvY = calculate_Chol_solve(self.Chol_factor, y_mean).reshape(y_mean.shape)
        self.logdet_KV = calculate_Chol_logdet(self.Chol_factor)

    def _new_KVlogdet_KVinvY(self, hps):
        """fvgp/gp_kv.py:574-593 (Chol branch) -- touches no state."""
        K = self.kernel(self.x_data, self.x_data, hps)
        V = self.noise(self.x_data)
        m = self.mean(self.x_data)
        KV = addKV(K, V)
        L = calculate_Chol_factor(KV)
        y_mean = self.y_data - m[:, None]
        return calculate_Chol_solve(L, y_mean).reshape(y_mean.shape), calculate_Chol_logdet(L), m, KV

    # --- log marginal likelihood --------------------------------------------------
    def log_likelihood(self, hyperparameters=None):
        """fvgp/gp_marginal_likelihood.py:137-179."""
        if hyperparameters is None:
            KVinvY, logdet, m = self.KVinvY, self.logdet_KV, self.m
        else:
            try:
                KVinvY, logdet, m, _ = self._new_KVlogdet_KVinvY(np.asarray(hyperparameters, dtype=float))
            except Exception as e:
                raise Exception(f"Linear algebra failed for hyperparameters {hyperparameters}: {e}") from e
        n = len(self.y_data)
        y_mean = self.y_data - m[:, None]
        l1 = np.sum(y_mean * KVinvY) / y_mean.shape[1]
        return -0.5 * (l1 + logdet + n * np.log(2.0 * np.pi))

    def neg_log_likelihood_gradient(self, hyperparameters=None, component=0):
        """fvgp/gp_marginal_likelihood.py:224-309, ram_economy False, default mean
        (dm/dh == 0, gp_prior.py:460-470) and measured noise (dV/dh == 0)."""
        if hyperparameters is None:
            hps, KVinvY, KV = self.hyperparameters, self.KVinvY, addKV(self.K, self.V)
        else:
            hps = np.asarray(hyperparameters, dtype=float)
            KVinvY, _, _, KV = self._new_KVlogdet_KVinvY(hps)
        b = KVinvY[:, component]
        dK = self.kernel_grad(self.x_data, self.x_data, hps)
        a = np.linalg.solve(np.array([KV] * len(hps)), dK)            # gp_lin_alg.py:1590
        bbT = np.outer(b, b.T)
        g = np.zeros(len(hps))
        for i in range(len(hps)):
            g[i] = -0.5 * (np.einsum('ij,ji->', bbT, dK[i]) - np.trace(a[i]))
        return g

    def neg_log_likelihood_gradient_potri(self, hyperparameters=None, component=0):
        """Same value by the N^3 route the HIP path takes:
        g_i = 1/2 ( sum_jk KVinv_jk dK_i,jk - b^T dK_i b )."""
        if hyperparameters is None:
            hps, KVinvY, L = self.hyperparameters, self.KVinvY, self.Chol_factor
        else:
            hps = np.asarray(hyperparameters, dtype=float)
            KVinvY, _, _, KV = self._new_KVlogdet_KVinvY(hps)
            L = calculate_Chol_factor(KV)
        b = KVinvY[:, component]
        KVinv = cho_solve((L, True), np.eye(len(b)))
        dK = self.kernel_grad(self.x_data, self.x_data, hps)
        return np.array([0.5 * (np.sum(KVinv * dK[i]) - b @ dK[i] @ b) for i in range(len(hps))])

    # --- posterior ------------------------------------------------------------------
    def posterior_mean(self, x_pred, hyperparameters=None, x_out=None):
        """fvgp/gp_posterior.py:139-182."""
        KVinvY = self.KVinvY
        hps = self.hyperparameters
        if hyperparameters is not None:
            hps = np.asarray(hyperparameters, dtype=float)
            KVinvY = self._new_KVlogdet_KVinvY(hps)[0]
        if x_out is None:
            x_out = self.x_out
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = cartesian_product(x_pred, x_out)
        k = self.kernel(self.x_data, x_pred, hps)
        A = np.asarray(k.T @ KVinvY)
        post = self.mean(x_pred)[:, None] + A
        post_re = post.reshape(len(x_orig), len(x_out), order='F') if isinstance(x_out, np.ndarray) else post
        if KVinvY.shape[1] == 1 and not isinstance(x_out, np.ndarray):
            return {"x": x_orig, "m(x)": np.squeeze(post_re), "m(x)_flat": np.squeeze(post), "x_pred": x_pred}
        if KVinvY.shape[1] == 1:
            return {"x": x_orig, "m(x)": post_re, "m(x)_flat": np.squeeze(post), "x_pred": x_pred}
        return {"x": x_orig, "m(x)": post_re, "m(x)_flat": post, "x_pred": x_pred}

    def posterior_covariance(self, x_pred, x_out=None, variance_only=False, add_noise=False):
        """fvgp/gp_posterior.py:229-288 (Chol mode: full S always formed, :246)."""
        if x_out is None:
            x_out = self.x_out
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = cartesian_product(x_pred, x_out)
        k = self.kernel(self.x_data, x_pred, self.hyperparameters)
        kk = self.kernel(x_pred, x_pred, self.hyperparameters)
        S = kk - k.T @ calculate_Chol_solve(self.Chol_factor, k)        # :120-136
        v = np.array(np.diag(S))
        if np.any(v < -0.0001):
            warnings.warn("Negative variances encountered. That normally means that the model is unstable. ")
        if np.any(v < 0.0):
            v[v < 0.0] = 0.0
            if not variance_only:
                np.fill_diagonal(S, v)
        if add_noise:
            # gp_posterior.py:554-569: the likelihood's noise function is always callable
            # (measured -> gp_likelihood.py:106-110, default -> :102-104), 1-d result
            noise = self.noise(x_pred)
            v = v + noise
            S = S + np.diag(noise)
        if isinstance(x_out, np.ndarray):
            v_re = v.reshape(len(x_orig), len(x_out), order='F')
            S_re = S.reshape(len(x_orig), len(x_out), len(x_orig), len(x_out), order='F').transpose(0, 2, 1, 3)
        else:
            v_re, S_re = v, S
            if self.y_data.shape[1] > 1:
                v = np.tile(v[:, None], (1, self.y_data.shape[1]))
                v_re = np.tile(v_re[:, None], (1, self.y_data.shape[1]))
        return {"x": x_orig, "x_pred": x_pred, "v(x)": v_re, "S": S_re, "S_flat": S, "v_flat": v}


    # -- derivatives built on the path (all finite-difference based in the reference) -------------------
    def _d_kernel_dx(self, x1, x2, direction, hps):
        """fvgp/gp_prior.py:402-409: forward difference, step 1e-8, in coordinate `direction` of x1."""
        new_points = np.array(x1)
        epsilon = 1e-8
        new_points[:, direction] += epsilon
        return (self.kernel(new_points, x2, hps) - self.kernel(x1, x2, hps)) / epsilon

    def posterior_mean_grad(self, x_pred, hyperparameters=None, x_out=None, direction=None, component=0):
        """fvgp/gp_posterior.py:184-226."""
        KVinvY = self.KVinvY[:, component]
        if hyperparameters is not None:
            KVinvY = self._new_KVlogdet_KVinvY(np.asarray(hyperparameters))[0][:, component]
        else:
            hyperparameters = self.hyperparameters
        if x_out is None:
            x_out = self.x_out
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = cartesian_product(x_pred, x_out)
        f = self._mean_at(x_pred, hyperparameters)
        eps = 1e-6

        def one(direction):
            x1 = np.array(x_pred)
            x1[:, direction] = x1[:, direction] + eps
            mean_der = (self._mean_at(x1, hyperparameters) - f) / eps
            k_g = self._d_kernel_dx(x_pred, self.x_data, direction, hyperparameters)
            return mean_der + (k_g @ KVinvY)

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

    def _mean_at(self, x, hps):
        return self.mean(x)            # default prior mean (gp_prior.py:449-458): independent of hps and of x

    def posterior_covariance_grad(self, x_pred, x_out=None, direction=None):
        """fvgp/gp_posterior.py:290-331."""
        if x_out is None:
            x_out = self.x_out
        x_orig = x_pred.copy()
        if isinstance(x_out, np.ndarray):
            x_pred = cartesian_product(x_pred, x_out)
        hps = self.hyperparameters
        k = self.kernel(self.x_data, x_pred, hps)
        k_covariance_prod = calculate_Chol_solve(self.Chol_factor, k)
        eps = 1e-6

        def dS(direction):
            k_g = self._d_kernel_dx(x_pred, self.x_data, direction, hps).T
            x1 = np.array(x_pred)
            x1[:, direction] = x1[:, direction] + eps
            kk_g = (self.kernel(x1, x1, hps) - self.kernel(x_pred, x_pred, hps)) / eps
            return kk_g - (2.0 * k_g.T @ k_covariance_prod)

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

    def neg_log_likelihood_hessian(self, hyperparameters):
        """fvgp/gp_marginal_likelihood.py:312-336: forward difference (1e-6) of the exact gradient, symmetrised."""
        nh = len(hyperparameters)
        d2 = np.zeros((nh, nh))
        epsilon = 1e-6
        g0 = self.neg_log_likelihood_gradient(hyperparameters=hyperparameters)
        for i in range(nh):
            t = np.array(hyperparameters)
            t[i] = t[i] + epsilon
            d2[i, i:] = ((self.neg_log_likelihood_gradient(hyperparameters=t) - g0) / epsilon)[i:]
        return d2 + d2.T - np.diag(np.diag(d2))

    def test_log_likelihood_gradient(self, hyperparameters, epsilon=1e-6):
        """fvgp/gp_marginal_likelihood.py:338-364: (forward-difference gradient, analytical gradient)."""
        thps = np.array(hyperparameters)
        grad = np.empty(len(thps))
        for i in range(len(thps)):
            aux = np.array(thps)
            aux[i] = aux[i] + epsilon
            grad[i] = (self.log_likelihood(hyperparameters=aux) - self.log_likelihood(hyperparameters=thps)) / epsilon
        return grad, -self.neg_log_likelihood_gradient(hyperparameters=thps)


# ---------------------------------------------------------------------------------------------
# validation scores and P x P information measures (fvgp/gp.py:1754-2071, gp_posterior.py:391-552),
# restated on OracleGP's posterior -- callers of the path, kept here so the facade's versions have a checker
# ---------------------------------------------------------------------------------------------
def validation_scores(o, x_test, y_test, interval=0.95):
    from scipy.stats import norm
    mean = o.posterior_mean(x_test)["m(x)"]
    v = o.posterior_covariance(x_test)["v(x)"]
    vn = o.posterior_covariance(x_test, add_noise=True)["v(x)"]
    out = {}
    out["rmse"] = np.sqrt(np.sum((y_test - mean) ** 2) / y_test.size)                      # gp.py:1784-1805
    out["nrmse"] = out["rmse"] / (np.max(y_test) - np.min(y_test))                         # :1807-1825
    out["mae"] = np.mean(np.abs(y_test - mean))                                            # :1994-2014
    out["mape"] = np.mean(np.abs((y_test - mean) / y_test))                                # :2016-2038
    out["r2"] = 1. - np.sum((y_test - mean) ** 2) / np.sum((y_test - np.mean(y_test)) ** 2)   # :1854-1874
    nl = np.mean(0.5 * np.log(2 * np.pi * v) + 0.5 * ((y_test - mean) ** 2) / v)           # :1827-1852
    out["nlpd"] = nl
    bm, bv = np.mean(o.y_data), np.var(o.y_data)                                           # :2040-2071
    out["msll"] = nl - np.mean(0.5 * np.log(2 * np.pi * bv) + 0.5 * ((y_test - bm) ** 2) / bv)
    sigma = np.sqrt(v)                                                                     # :1754-1782
    res = abs(sigma * ((1. / np.sqrt(np.pi)) - 2. * norm.pdf((y_test - mean) / sigma)
                       - (((y_test - mean) / sigma) * (2. * norm.cdf((y_test - mean) / sigma) - 1.))))
    out["crps_mean"], out["crps_std"] = np.mean(res), np.sqrt(np.var(res))
    sn = np.sqrt(vn)                                                                       # :1876-1992
    z = norm.ppf(1 - (1 - interval) / 2)
    lo, hi = mean - z * sn, mean + z * sn
    out["picp"] = np.mean((y_test >= lo) & (y_test <= hi))
    out["mpiw"] = np.mean(2 * z * np.sqrt(np.clip(vn, 0.0, None)))
    a = 1 - interval
    out["interval_score"] = np.mean((hi - lo) + (2 / a) * np.maximum(lo - y_test, 0) + (2 / a) * np.maximum(y_test - hi, 0))
    return out


def kl_div(mu1, mu2, S1, S2):
    """gp_posterior.py:408-424."""
    ld1 = np.linalg.slogdet(S1)[1]
    ld2 = np.linalg.slogdet(S2)[1]
    x1 = np.linalg.solve(S2, S1)
    mu = np.subtract(mu2, mu1)
    x2 = np.linalg.solve(S2, mu)
    return abs(0.5 * (np.trace(x1) + x2 @ mu - float(len(mu)) + (ld2 - ld1)))


def information_measures(o, x_pred, comp_mean, comp_cov):
    """gp_kl_div, gp_relative_information_entropy(_set), posterior_probability (gp_posterior.py:426-440,494-552)."""
    pm = o.posterior_mean(x_pred)["m(x)_flat"]
    S = o.posterior_covariance(x_pred)["S_flat"]
    eye = np.identity(len(S))
    out = {"kl_div": kl_div(pm, comp_mean, S + eye * 1e-9, comp_cov + eye * 1e-9)}
    kk = o.kernel(x_pred, x_pred, o.hyperparameters) + eye * 1e-9
    out["rie"] = kl_div(o.mean(x_pred), pm, kk, S + eye * 1e-9)
    rie_set = []
    for i in range(len(x_pred)):
        xi = x_pred[i].reshape(1, -1)
        rie_set.append(kl_div(o.mean(xi), o.posterior_mean(xi)["m(x)_flat"], o.kernel(xi, xi, o.hyperparameters) + 1e-9,
                              o.posterior_covariance(xi)["S_flat"] + 1e-9))
    out["rie_set"] = np.array(rie_set)
    gcov = o.posterior_covariance(x_pred, add_noise=True)["S_flat"]
    gi, ci = np.linalg.inv(gcov), np.linalg.inv(comp_cov)
    cov = np.linalg.inv(gi + ci)
    mu = cov @ gi @ pm + cov @ ci @ comp_mean
    C = 0.5 * (((pm.T @ gi + comp_mean.T @ ci).T @ cov @ (gi @ pm + ci @ comp_mean))
               - (pm.T @ gi @ pm + comp_mean.T @ ci @ comp_mean)).squeeze()
    ln_p = (C + 0.5 * np.linalg.slogdet(cov)[1]) - (np.log((2.0 * np.pi) ** (len(mu) / 2.0))
                                                    + 0.5 * (np.linalg.slogdet(gcov)[1] + np.linalg.slogdet(comp_cov)[1]))
    out["pp_mu"], out["pp_cov"], out["pp_prob"] = mu, cov, np.exp(ln_p)
    return out


def log_likelihood_once(x, y, noise_variances, hps, kernel="rbf_ard"):
    """One metric unit on the CPU: K-assembly + addKV + potrf + potrs + logdet + scalar.
    Used as bench.py's cpu_baseline ("port").  Returns (value, dict of stage seconds)."""
    import time
    kfun = KERNELS[kernel]
    t0 = time.perf_counter()
    K = kfun(x, x, hps)
    t1 = time.perf_counter()
    KV = addKV(K, noise_variances)
    del K
    t2 = time.perf_counter()
    L = calculate_Chol_factor(KV)
    t3 = time.perf_counter()
    y2 = y.reshape(len(y), -1)
    m = np.mean(y2)
    ym = y2 - m
    a = calculate_Chol_solve(L, ym)
    ld = calculate_Chol_logdet(L)
    val = -0.5 * (np.sum(ym * a) / ym.shape[1] + ld + len(y2) * np.log(2.0 * np.pi))
    t4 = time.perf_counter()
    return float(val), {"kmat": t1 - t0, "addKV": t2 - t1, "potrf": t3 - t2, "solve_logdet": t4 - t3}
