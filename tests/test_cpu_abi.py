"""The CPU twin of the C ABI (oracle/cpu_abi/fvgp_cpu.c -> oracle/_cpu/libfvgp_cpu.so; test infrastructure) held to the
reference's golden outputs, and -- where the reference tree is present (the build container) -- DRIVEN BY THE REFERENCE
ITSELF through its own plug points: kernel_function and linalg_mode=[f_factor, f_solve, f_logdet]
(fvgp/gp_kv.py:457-458,552-554,625-628,697-715; tests/test_fvgp.py:417-426).  That run validates the ABI's semantics
(padding, lower-triangle-only reads, 1-d -> (n,1) reshape of the right-hand side, info / return codes) end to end
with the reference's control flow on top, without a GPU."""
import ctypes
import os
import subprocess
import warnings

import numpy as np
import pytest

from conftest import ROOT, load_golden

LIB = os.environ.get("FVGP_CPU_LIB") or os.path.join(ROOT, "oracle", "_cpu", "libfvgp_cpu.so")     # (make -C oracle asan-test: the sanitizer build)
REF = "/root/reference"
c_i, c_l, c_d, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
KID = {"rbf_ard": 0, "matern32_ard": 1, "matern52_ard": 2}


def pad128(n):
    return (n + 127) // 128 * 128


@pytest.fixture(scope="module")
def cpu():
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    L = ctypes.CDLL(LIB)
    L.fvgp_hip_last_error_string.restype = ctypes.c_char_p
    h = c_p()
    assert L.fvgp_hip_create(ctypes.byref(h), 0, None) == 0
    return L, h


def ptr(a):
    return a.ctypes.data_as(c_p)


def dbl(a):
    return a.ctypes.data_as(ctypes.POINTER(c_d))


def kmat(cpu, kid, x1, x2, theta, vdiag=None, uplo=0, pad=0, fill=0.0):
    L, h = cpu
    r, c = (pad128(len(x1)), pad128(len(x2))) if pad else (len(x1), len(x2) + (len(x2) & 1))
    K = np.full((r, c), fill)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    rc = L.fvgp_hip_kmat(h, kid, ptr(x1), c_l(len(x1)), ptr(x2), c_l(len(x2)), x1.shape[1], dbl(th), len(th),
                         ptr(vdiag) if vdiag is not None else None, ptr(K), c_l(c), uplo, pad)
    assert rc == 0, (rc, L.fvgp_hip_last_error_string())
    return K


def test_cpu_twin_exports_the_abi_subset(cpu):
    L, _ = cpu
    for s in ("version", "last_error_string", "padded_dim", "workspace_bytes", "create", "destroy", "sync", "set_option",
              "get_profile", "invalidate_factor", "kmat", "potrf", "potrs", "trsm_lower", "trsm_lower_t", "logdet", "potri",
              "loglik", "loglik_grad", "grad_trace", "posterior", "gemm", "add_lower", "symmetrize"):
        assert hasattr(L, "fvgp_hip_" + s), s
    assert L.fvgp_hip_version() == 100


@pytest.mark.parametrize("name", ["G2_rbf_n512_d3.npz", "G3_matern52_n512_d3.npz", "G6_rbf_2col_n300_d3.npz"])
def test_cpu_twin_matches_the_reference_vectors(cpu, name):
    """fused log-likelihood, gradient and posterior through the twin's C entry points against the golden outputs; the
    strict upper triangle of the scratch matrix is poisoned with NaN (never read, never written: cho_factor's contract)"""
    L, h = cpu
    fx = load_golden(name)
    x, theta, nv = np.ascontiguousarray(fx["x"]), fx["theta"], np.ascontiguousarray(fx["noise_variances"])
    y = fx["y"].reshape(len(x), -1)
    n, d = x.shape
    ncol = y.shape[1]
    kid = KID[str(fx["kernel"])]
    npad = pad128(n)
    ym = np.ascontiguousarray(y - np.mean(y))
    KV = np.full((npad, npad), np.nan)
    KV[np.tril_indices(npad)] = 0.0
    alpha = np.zeros((npad, ncol))
    out = (c_d * 3)()
    info = c_i(-1)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    rc = L.fvgp_hip_loglik(h, kid, ptr(x), c_l(n), d, dbl(th), len(th), ptr(nv), ptr(ym), ncol, ptr(KV), c_l(npad), ptr(alpha), out, ctypes.byref(info))
    assert rc == 0 and info.value == 0
    np.testing.assert_allclose(out[0], fx["loglik"], rtol=1e-11)
    np.testing.assert_allclose(out[1], fx["logdet"], rtol=1e-11)
    assert np.max(np.abs(alpha[:n] - fx["KVinvY"])) <= 1e-9 * np.max(np.abs(fx["KVinvY"]))
    np.testing.assert_allclose(np.diag(KV)[:n], fx["L_diag"], rtol=1e-11)
    assert np.all(np.isnan(KV[np.triu_indices(npad, 128)]))                 # tiles above the diagonal: untouched
    assert np.all(np.diag(KV)[n:] == 1.0)                                   # identity padding
    # posterior from the factor
    xp = np.ascontiguousarray(fx["x_pred"])
    P, Pp = len(xp), pad128(len(xp))
    kx = np.zeros((npad, Pp)); mean = np.zeros((P, ncol)); var = np.zeros(P); S = np.zeros((Pp, Pp))
    rc = L.fvgp_hip_posterior(h, kid, ptr(x), c_l(n), d, dbl(th), len(th), ptr(KV), c_l(npad), ptr(alpha), ncol, ptr(xp), c_l(P),
                              ptr(kx), c_l(Pp), ptr(mean), ptr(var), ptr(S), c_l(Pp))
    assert rc == 0
    np.testing.assert_allclose(np.squeeze(mean + np.mean(y)), fx["pm"], rtol=1e-9, atol=1e-11)
    assert np.max(np.abs(S[:P, :P] - fx["pS"])) <= 1e-10 * theta[0]
    assert np.max(np.abs(var - np.diag(fx["pS"]))) <= 1e-10 * theta[0]
    # gradient (destroys the factor: KV^-1 lower on return)
    work = np.zeros((npad, npad))
    g = (c_d * len(th))()
    rc = L.fvgp_hip_loglik_grad(h, kid, ptr(x), c_l(n), d, dbl(th), len(th), ptr(alpha), ncol, 0, ptr(KV), c_l(npad), ptr(work), c_l(npad), g)
    assert rc == 0
    np.testing.assert_allclose(np.array(g[:]), fx["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))


def test_cpu_twin_error_codes(cpu):
    L, h = cpu
    fx = load_golden("G7_nonpd.npz")
    M = np.ascontiguousarray(fx["M"])
    n = len(M)
    A = np.zeros((pad128(n), pad128(n)))
    A[:n, :n] = M
    info = c_i(0)
    assert L.fvgp_hip_potrf(h, ptr(A), c_l(n), c_l(pad128(n)), ctypes.byref(info)) == 0
    assert info.value == int(fx["info"])                                    # dpotrf's info: first non-positive leading minor
    assert L.fvgp_hip_potrf(None, ptr(A), c_l(n), c_l(pad128(n)), ctypes.byref(info)) == -1
    assert L.fvgp_hip_potrf(h, ptr(A), c_l(n), c_l(n - 1), ctypes.byref(info)) == -4
    out = (c_d * 3)()
    x = np.zeros((4, 2)); th = np.ones(3)
    assert L.fvgp_hip_loglik(h, 0, ptr(x), c_l(4), 2, dbl(th), 3, None, ptr(x), 1, ptr(A), c_l(128), ptr(x), out, None) == -8


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fvgp")), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("name", ["G2_rbf_n512_d3.npz", "G3_matern52_n512_d3.npz"])
def test_the_reference_drives_the_abi_through_its_plug_points(cpu, name):
    """The imported reference GP with kernel_function = the twin's kmat and linalg_mode = [potrf, potrs, logdet] of the
    twin: its own control flow (state build, log_likelihood at a new theta, posterior) on top of the ABI reproduces its
    own default-path outputs (the golden vectors)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import import_reference
    fvgp = import_reference()
    L, h = cpu
    fx = load_golden(name)
    kid = KID[str(fx["kernel"])]
    calls = {"factor": 0, "solve": 0, "logdet": 0, "kernel": 0}

    def kernel(x1, x2, hps):
        calls["kernel"] += 1
        x1, x2 = np.ascontiguousarray(x1, dtype=np.float64), np.ascontiguousarray(x2, dtype=np.float64)
        return kmat(cpu, kid, x1, x2, hps)[:, :len(x2)].copy()

    def f_factor(KV):
        calls["factor"] += 1
        assert isinstance(KV, np.ndarray) and KV.dtype == np.float64 and KV.ndim == 2
        n = len(KV)
        A = np.full((pad128(n), pad128(n)), np.nan)
        A[np.tril_indices(pad128(n))] = 0.0
        A[:n, :n][np.tril_indices(n)] = KV[np.tril_indices(n)]               # only the lower triangle crosses the boundary
        info = c_i(0)
        assert L.fvgp_hip_potrf(h, ptr(A), c_l(n), c_l(pad128(n)), ctypes.byref(info)) == 0
        if info.value:
            raise np.linalg.LinAlgError(f"{info.value}-th leading minor of the array is not positive definite")
        return A, n

    def f_solve(obj, b):
        calls["solve"] += 1
        A, n = obj
        b2 = np.asarray(b, dtype=np.float64).reshape(n, -1)                   # 1-d -> (n, 1): gp_lin_alg.py:292
        B = np.zeros((pad128(n), b2.shape[1]))
        B[:n] = b2
        assert L.fvgp_hip_potrs(h, ptr(A), c_l(n), c_l(pad128(n)), ptr(B), c_l(b2.shape[1]), c_l(b2.shape[1])) == 0
        return B[:n].copy()

    def f_logdet(obj):
        calls["logdet"] += 1
        A, n = obj
        out = c_d(0.0)
        assert L.fvgp_hip_logdet(h, ptr(A), c_l(n), c_l(pad128(n)), ctypes.byref(out)) == 0
        return out.value

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                     kernel_function=kernel, linalg_mode=[f_factor, f_solve, f_logdet])
        np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-11)
        for t, ll in zip(fx["thetas"], fx["logliks"]):
            np.testing.assert_allclose(gp.log_likelihood(t), ll, rtol=1e-11)
        assert np.max(np.abs(gp.kv.KVinvY - fx["KVinvY"])) <= 1e-9 * np.max(np.abs(fx["KVinvY"]))
        np.testing.assert_allclose(gp.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-9, atol=1e-11)
        pc = gp.posterior_covariance(fx["x_pred"])
        assert np.max(np.abs(pc["S"] - fx["pS"])) <= 1e-10 * fx["theta"][0]
        assert np.max(np.abs(pc["v(x)"] - fx["pv"])) <= 1e-10 * fx["theta"][0]
    assert min(calls.values()) >= 1, calls
