"""fvgp_amd.GP / fvGP against the reference's own outputs (tests/golden) -- reads like the reference's
tests: build a GP, call the public methods, compare.  Tolerances: SURVEY 8c / BASELINE.md section 5."""
import pickle
import warnings

import numpy as np
import pytest

from conftest import load_golden
from oracle import fvgp_oracle as orc

pytestmark = pytest.mark.gpu

CASES = [("G1_rbf_n500_d1.npz", {}), ("G2_rbf_n512_d3.npz", {}), ("G3_matern52_n512_d3.npz", {}),
         ("G4_default_n256_d2.npz", {"default": True}), ("G6_rbf_2col_n300_d3.npz", {})]


def _make(fx, opt):
    import fvgp_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if opt.get("default"):
            return fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"])
        return fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                           kernel_function=str(fx["kernel"]))


@pytest.mark.parametrize("name,opt", CASES)
def test_gp_matches_reference(name, opt):
    fx = load_golden(name)
    gp = _make(fx, opt)
    sig = fx["theta"][0]
    # state
    assert np.max(np.abs(gp.K[:8, :8] - fx["K_corner"])) <= 8e-16 * sig
    np.testing.assert_allclose(gp.V, fx["V"], rtol=1e-15)
    np.testing.assert_allclose(gp.m, fx["m"], rtol=1e-15)
    np.testing.assert_allclose(np.diag(gp.Chol_factor), fx["L_diag"], rtol=1e-10)
    assert np.max(np.abs(gp.KVinvY - fx["KVinvY"])) <= 1e-8 * np.max(np.abs(fx["KVinvY"]))
    # likelihood: cached, explicit theta, sweep
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(fx["theta"]), fx["loglik_theta"], rtol=1e-10)
    np.testing.assert_allclose(gp.neg_log_likelihood(fx["thetas"][0]), -fx["logliks"][0], rtol=1e-10)
    for t, ll in zip(fx["thetas"], fx["logliks"]):
        np.testing.assert_allclose(gp.log_likelihood(t), ll, rtol=1e-10)
    # an evaluation at another theta must not have touched the state (gp_kv.py:574-578)
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
    # gradient
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"]), fx["grad"], rtol=1e-8,
                               atol=1e-9 * np.max(np.abs(fx["grad"])))
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(), fx["grad_cached"], rtol=1e-8,
                               atol=1e-9 * np.max(np.abs(fx["grad"])))
    if "grad_c1" in fx:
        np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"], component=1), fx["grad_c1"], rtol=1e-8,
                                   atol=1e-9 * np.max(np.abs(fx["grad_c1"])))
    # posterior
    xp = fx["x_pred"]
    pm = gp.posterior_mean(xp)
    assert set(pm) == {"x", "m(x)", "m(x)_flat", "x_pred"}
    assert pm["m(x)"].shape == fx["pm"].shape and pm["m(x)_flat"].shape == fx["pm_flat"].shape
    np.testing.assert_allclose(pm["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gp.posterior_mean(xp, hyperparameters=fx["thetas"][0])["m(x)"], fx["pm_theta1"],
                               rtol=1e-8, atol=1e-10)
    pc = gp.posterior_covariance(xp)
    pcn = gp.posterior_covariance(xp, add_noise=True)
    assert set(pc) == {"x", "x_pred", "v(x)", "S", "S_flat", "v_flat"}
    for key, tag in (("v(x)", "pv"), ("S", "pS"), ("S_flat", "pS_flat"), ("v_flat", "pv_flat")):
        assert np.asarray(pc[key]).shape == fx[tag].shape
        assert np.max(np.abs(pc[key] - fx[tag])) <= 1e-10 * sig + 1e-12
        assert np.max(np.abs(pcn[key] - fx[tag + "_noise"])) <= 1e-10 * sig + 1e-12
    pv = gp.posterior_covariance(xp, variance_only=True)
    assert np.max(np.abs(pv["v(x)"] - fx["pv"])) <= 1e-10 * sig + 1e-12


@pytest.mark.parametrize("name", ["G5_fvgp_4x64.npz", "G5n_fvgp_4x64_nan.npz"])
def test_fvgp_multitask_matches_reference(name):
    """MultiTaskTest shape: (V,Di) x (V,No) -> task-major index set; posterior reshapes of
    gp_posterior.py:264-274 (pinned by tests/test_fvgp.py:1973-2012)."""
    import fvgp_amd
    fx = load_golden(name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.fvGP(fx["fvgp_x"], fx["fvgp_y"], init_hyperparameters=fx["theta"], noise_variances=fx["fvgp_noise"])
    assert np.array_equal(gp.x_data, fx["x"])
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
    for t, ll in zip(fx["thetas"], fx["logliks"]):
        np.testing.assert_allclose(gp.log_likelihood(t), ll, rtol=1e-10)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"]), fx["grad"], rtol=1e-8,
                               atol=1e-9 * np.max(np.abs(fx["grad"])))
    xp = fx["x_pred"]
    for kw in ({}, {"x_out": fx["x_out"]}):
        pm = gp.posterior_mean(xp, **kw)
        assert pm["m(x)"].shape == (len(xp), 4)
        np.testing.assert_allclose(pm["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
        assert np.array_equal(pm["x_pred"], fx["pm_xpred"])
        pc = gp.posterior_covariance(xp, **kw)
        assert pc["S"].shape == (len(xp), len(xp), 4, 4) and pc["v(x)"].shape == (len(xp), 4)
        assert np.max(np.abs(pc["S"] - fx["pS"])) <= 1e-10 and np.max(np.abs(pc["v(x)"] - fx["pv"])) <= 1e-10


def test_nonpd_is_reported_like_the_reference():
    """tests/test_fvgp.py:4653-4665,3738-3768: NonPositiveDefiniteError(LinAlgError) with diagnostics, and
    log_likelihood(theta) re-raises as 'Linear algebra failed for hyperparameters ...'."""
    import fvgp_amd
    from fvgp_amd import gp_lin_alg
    fx = load_golden("G7_nonpd.npz")
    with pytest.raises(fvgp_amd.NonPositiveDefiniteError) as ei:
        gp_lin_alg.calculate_Chol_factor(fx["M"])
    assert isinstance(ei.value, np.linalg.LinAlgError)
    assert f"the 96x96 prior covariance matrix is not positive definite" in str(ei.value)
    assert f"{int(fx['info'])}-th leading minor" in str(ei.value)
    f = gp_lin_alg.calculate_Chol_factor(fx["Mok"])
    np.testing.assert_allclose(f.lower(), fx["Lok"], rtol=0, atol=1e-13 * np.max(fx["Lok"]))
    np.testing.assert_allclose(gp_lin_alg.calculate_Chol_solve(f, fx["rhs"]), fx["sol"], rtol=1e-11, atol=1e-14)
    assert gp_lin_alg.calculate_Chol_solve(f, fx["rhs"][:, 0]).shape == (96, 1)
    np.testing.assert_allclose(gp_lin_alg.calculate_Chol_logdet(f), float(fx["logdet"]), rtol=1e-13)
    # duplicate points + negligible noise: a numerically singular K inside GP.log_likelihood(theta)
    rng = np.random.default_rng(0)
    x = rng.random((200, 2)); x[150:] = x[:50]
    y = np.sin(x.sum(axis=1))
    gp = fvgp_amd.GP(x, y, init_hyperparameters=np.array([1.0, 5.0, 5.0]), noise_variances=np.full(200, 1e-2),
                     kernel_function="rbf_ard")

    def tiny_noise(xx, h):
        return np.full(len(xx), 1e-300)
    gp2 = fvgp_amd.GP(x, y, init_hyperparameters=np.array([1.0, 0.3, 0.3]), noise_function=lambda xx, h: np.full(len(xx), 1e-2),
                      kernel_function="rbf_ard")
    gp2._noise_callable = tiny_noise
    with pytest.raises(Exception, match="Linear algebra failed for hyperparameters"):
        gp2.log_likelihood(np.array([1.0, 50.0, 50.0]))
    assert np.isfinite(gp.log_likelihood())


def test_host_callable_kernel_slow_path_and_custom_mean_noise():
    """An arbitrary Python kernel / mean / noise callable keeps working (gp.py:80-101): K comes from the
    host, everything after addKV runs on the device."""
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    x, y, th = fx["x"], fx["y"], fx["theta"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=fx["noise_variances"],
                         kernel_function=orc.rbf_ard, kernel_function_grad=orc.rbf_ard_grad)
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(fx["thetas"][1]), fx["logliks"][1], rtol=1e-10)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(th), fx["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))
    np.testing.assert_allclose(gp.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(gp.posterior_covariance(fx["x_pred"])["S"] - fx["pS"])) <= 1e-10
    # hyperparameter-dependent mean and noise: theta = [sig, l1, l2, l3, mean level, noise level]
    mean = lambda xx, h: np.full(len(xx), h[4])
    noise = lambda xx, h: np.full(len(xx), h[5])
    th6 = np.concatenate([th, [0.1, 0.02]])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th6, kernel_function="rbf_ard", prior_mean_function=mean, noise_function=noise)
    K = orc.rbf_ard(x, x, th6[:4]); KV = orc.addKV(K, noise(x, th6))
    Lr = orc.calculate_Chol_factor(KV); ym = (y - mean(x, th6)).reshape(-1, 1)
    ref = -0.5 * (np.sum(ym * orc.calculate_Chol_solve(Lr, ym)) + orc.calculate_Chol_logdet(Lr) + len(y) * np.log(2 * np.pi))
    np.testing.assert_allclose(gp.log_likelihood(), ref, rtol=1e-10)
    g = gp.neg_log_likelihood_gradient(th6)
    fd = np.zeros(6)
    for i in range(6):
        e = np.zeros(6); e[i] = 1e-6 * max(1.0, abs(th6[i]))
        fd[i] = -(gp.log_likelihood(th6 + e) - gp.log_likelihood(th6 - e)) / (2 * e[i])
    np.testing.assert_allclose(g, fd, rtol=2e-5, atol=1e-4)


def test_callable_kernel_without_gradient_takes_the_reference_finite_differences():
    """SURVEY 8 row a13, third route: a kernel callable with no kernel_function_grad is differentiated by central
    differences with eps = 1e-8 (gp_prior.py:438-447; one direction at a time under ram_economy, :236-240).  G1 holds the
    gradient the REFERENCE computes this way; the host callable here is the oracle's expression-for-expression kernel, so
    the difference matrices are the reference's and only KV^-1 and the trace come from the device."""
    import fvgp_amd
    fx = load_golden("G1_rbf_n500_d1.npz")
    x, y, th, nv = fx["x"], fx["y"], fx["theta"], fx["noise_variances"]
    calls = []

    def grad_dir(x1, x2, h, direction):                 # the ram_economy signature of kernel_function_grad
        calls.append(direction)
        return orc.rbf_ard_grad(x1, x2, h)[direction]

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=lambda a, b, h: orc.rbf_ard(a, b, h))
        gp_re = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=lambda a, b, h: orc.rbf_ard(a, b, h),
                            ram_economy=True)
        gp_dir = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=lambda a, b, h, args: orc.rbf_ard(a, b, h),
                             kernel_function_grad=grad_dir, ram_economy=True)
    g = gp.neg_log_likelihood_gradient(th)
    # the reference's own finite-difference result; the noise of an eps = 1e-8 difference quotient is ~1e-8 per entry of dK
    np.testing.assert_allclose(g, fx["grad_fd"], rtol=1e-6)
    np.testing.assert_allclose(gp_re.neg_log_likelihood_gradient(th), g, rtol=1e-12)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(), g, rtol=1e-9)       # hyperparameters=None: the state
    # user gradient, one direction per call (4-argument kernel: args is passed through)
    np.testing.assert_allclose(gp_dir.neg_log_likelihood_gradient(th), fx["grad"], rtol=1e-8)
    assert calls == [0, 1]


def test_linalg_mode_callables_and_inv_on_the_facade():
    """gp_kv.py:138-147: linalg_mode = [f_factor, f_solve, f_logdet] hands K+V to the user's host callables (reference tests
    tests/test_fvgp.py:417-426,5030-5052: the callables are really invoked and reproduce the Chol results); "Inv" keeps an
    explicit inverse like "CholInv"."""
    import fvgp_amd
    from scipy.linalg import cho_factor, cho_solve
    fx = load_golden("G2_rbf_n512_d3.npz")
    x, y, th, nv = fx["x"], fx["y"], fx["theta"], fx["noise_variances"]
    calls = {"f": 0, "s": 0, "l": 0}

    def f_factor(KV):
        calls["f"] += 1
        assert isinstance(KV, np.ndarray) and KV.shape == (len(x), len(x)) and np.allclose(KV, KV.T)
        return cho_factor(KV, lower=True)

    def f_solve(obj, b):
        calls["s"] += 1
        assert np.ndim(b) == 2
        return cho_solve(obj, b)

    def f_logdet(obj):
        calls["l"] += 1
        return 2.0 * np.sum(np.log(np.diag(obj[0])))

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard",
                         linalg_mode=[f_factor, f_solve, f_logdet])
        gpi = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard", linalg_mode="Inv")
    assert calls == {"f": 1, "s": 1, "l": 1}
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(fx["thetas"][0]), fx["logliks"][0], rtol=1e-10)
    assert calls["f"] == 2
    assert np.max(np.abs(gp.KVinvY - fx["KVinvY"])) <= 1e-8 * np.max(np.abs(fx["KVinvY"]))
    np.testing.assert_allclose(gp.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(gp.posterior_covariance(fx["x_pred"])["S"] - fx["pS"])) <= 1e-10
    assert calls["s"] >= 3
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(th), fx["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))
    # posterior_covariance_grad solves through f_solve too (the device buffer holds K + V in this mode, not a factor), and there is
    # no Cholesky factor to hand out (gp_kv.py:123)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gpc = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    xs = fx["x_pred"][:5]
    np.testing.assert_allclose(gp.posterior_covariance_grad(xs, direction=1)["dS/dx"],
                               gpc.posterior_covariance_grad(xs, direction=1)["dS/dx"], rtol=0, atol=2e-6)
    assert gp.Chol_factor is None and gpc.Chol_factor is not None
    gp.set_hyperparameters(fx["thetas"][1])
    np.testing.assert_allclose(gp.log_likelihood(), fx["logliks"][1], rtol=1e-10)
    # "Inv": explicit inverse, variance-only fast path
    np.testing.assert_allclose(gpi.log_likelihood(), fx["loglik"], rtol=1e-10)
    v = gpi.posterior_covariance(fx["x_pred"], variance_only=True)["v(x)"]
    assert np.max(np.abs(v - np.diag(fx["pS"]))) <= 1e-9


def test_matrix_valued_noise_model():
    """A noise function that returns a 2-d matrix: KV = K + V (gp_kv.py:654-657), gradient with the 3-d noise derivative
    (gp_marginal_likelihood.py:262-267), add_noise with the matrix at the prediction points (gp_posterior.py:554-569)."""
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    x, y, th = fx["x"][:300], fx["y"][:300], fx["theta"]

    def noise(xx, h):                                   # correlated noise: h[4] * (I + 0.3 exp(-|dx|^2 / 0.02))
        d2 = ((xx[:, None, :] - xx[None, :, :]) ** 2).sum(axis=2)
        return h[4] * (np.eye(len(xx)) + 0.3 * np.exp(-d2 / 0.02))

    th5 = np.concatenate([th, [0.02]])
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th5, kernel_function="rbf_ard", noise_function=noise)
    K = orc.rbf_ard(x, x, th5[:4])
    KV = orc.addKV(K, noise(x, th5))
    Lr = orc.calculate_Chol_factor(KV)
    ym = (y - np.mean(y)).reshape(-1, 1)
    a = orc.calculate_Chol_solve(Lr, ym)
    ref = -0.5 * (np.sum(ym * a) + orc.calculate_Chol_logdet(Lr) + len(y) * np.log(2 * np.pi))
    np.testing.assert_allclose(gp.log_likelihood(), ref, rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(th5), ref, rtol=1e-10)
    assert gp.V.shape == (300, 300)
    assert np.max(np.abs(gp.KVinvY - a)) <= 1e-8 * np.max(np.abs(a))
    # gradient: kernel part + 1/2 (tr(KV^-1 dV) - b^T dV b) for the noise level, against the dense formula
    g = gp.neg_log_likelihood_gradient(th5)
    KVinv = np.linalg.inv(KV)
    b = a[:, 0]
    dK = orc.rbf_ard_grad(x, x, th5[:4])
    dV = noise(x, np.array([0, 0, 0, 0, 1.0]))
    g_ref = np.array([0.5 * (np.sum(KVinv * d) - b @ d @ b) for d in list(dK) + [dV]])
    np.testing.assert_allclose(g, g_ref, rtol=2e-6, atol=1e-6 * np.max(np.abs(g_ref)))
    # posterior with the matrix-valued noise added at the prediction points
    xp = fx["x_pred"]
    k = orc.rbf_ard(x, xp, th5[:4])
    S_ref = orc.rbf_ard(xp, xp, th5[:4]) - k.T @ KVinv @ k + noise(xp, th5)
    pc = gp.posterior_covariance(xp, add_noise=True)
    assert np.max(np.abs(pc["S"] - S_ref)) <= 1e-9
    assert np.max(np.abs(pc["v(x)"] - np.diag(S_ref))) <= 1e-9


def test_pickle_roundtrip_and_update():
    """tests/test_fvgp.py:1108-1244: an unpickled GP answers posterior queries from the pickled factor."""
    import fvgp_amd
    fx = load_golden("G3_matern52_n512_d3.npz")
    gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                     kernel_function="matern52_ard")
    gp2 = pickle.loads(pickle.dumps(gp))
    np.testing.assert_allclose(gp2.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(gp2.posterior_covariance(fx["x_pred"])["S"] - fx["pS"])) <= 1e-10
    np.testing.assert_allclose(gp2.log_likelihood(), fx["loglik"], rtol=1e-10)
    # append == rebuild on the union
    x, y, nv = fx["x"], fx["y"], fx["noise_variances"]
    a = fvgp_amd.GP(x[:400], y[:400], init_hyperparameters=fx["theta"], noise_variances=nv[:400], kernel_function="matern52_ard")
    a.update_gp_data(x[400:], y[400:], noise_variances_new=nv[400:], append=True)
    np.testing.assert_allclose(a.log_likelihood(), fx["loglik"], rtol=1e-10)
    a.set_hyperparameters(fx["thetas"][2])
    np.testing.assert_allclose(a.log_likelihood(), fx["logliks"][2], rtol=1e-10)


def test_unpickled_factor_never_meets_another_factors_block_inverses():
    """The handle caches the inverted diagonal blocks of the last factor keyed on its address; torch's allocator
    hands a freed N x N block back at the same address.  A GP unpickled into the address of a freed, different,
    same-size GP must still answer from ITS factor (fvgp_hip_invalidate_factor)."""
    import gc
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    kw = dict(noise_variances=fx["noise_variances"], kernel_function="rbf_ard")
    keep = pickle.dumps(fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], **kw))
    gc.collect()
    for _ in range(3):                        # several rounds: whichever block the allocator returns, the answer stands
        other = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"] * np.array([3.0, 0.4, 2.0, 0.7]), **kw)
        other.posterior_covariance(fx["x_pred"])          # leaves ITS block inverses cached under its factor's address
        ptr = other._L.data_ptr()
        del other
        gc.collect()
        gp2 = pickle.loads(keep)
        same_block = gp2._L.data_ptr() == ptr
        assert np.max(np.abs(gp2.posterior_covariance(fx["x_pred"])["S"] - fx["pS"])) <= 1e-10, same_block
        np.testing.assert_allclose(gp2.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
        del gp2
        gc.collect()


def test_rank_n_append_equals_refactorisation():
    """update_gp_data(append=True): bordering update of the device factor (gp_lin_alg.py:1310-1477) against a
    from-scratch factorisation and the reference vectors; unaligned sizes on both sides of the 128 blocks."""
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    x, y, nv, th = fx["x"], fx["y"], fx["noise_variances"], fx["theta"]
    for n0 in (475, 384, 130):
        gp = fvgp_amd.GP(x[:n0], y[:n0], init_hyperparameters=th, noise_variances=nv[:n0], kernel_function="rbf_ard")
        gp.update_gp_data(x[n0:500], y[n0:500], noise_variances_new=nv[n0:500])          # append, rank-n
        gp.update_gp_data(x[500:], y[500:], noise_variances_new=nv[500:])                # a second, small append
        assert gp.point_number == 512
        np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-10)
        np.testing.assert_allclose(np.diag(gp.Chol_factor), fx["L_diag"], rtol=1e-10)
        assert np.max(np.abs(gp.KVinvY - fx["KVinvY"])) <= 1e-8 * np.max(np.abs(fx["KVinvY"]))
        np.testing.assert_allclose(gp.posterior_mean(fx["x_pred"])["m(x)"], fx["pm"], rtol=1e-8, atol=1e-10)
        assert np.max(np.abs(gp.posterior_covariance(fx["x_pred"])["S"] - fx["pS"])) <= 1e-10
        np.testing.assert_allclose(gp.log_likelihood(fx["thetas"][0]), fx["logliks"][0], rtol=1e-10)
    full = fvgp_amd.GP(x[:300], y[:300], init_hyperparameters=th, noise_variances=nv[:300], kernel_function="rbf_ard")
    full.update_gp_data(x[300:], y[300:], noise_variances_new=nv[300:], rank_n_update=False)
    np.testing.assert_allclose(full.log_likelihood(), fx["loglik"], rtol=1e-10)
    # an appended block whose Schur complement is not positive definite (negative "noise" on the new rows)
    bad = fvgp_amd.GP(x[:200], y[:200], init_hyperparameters=th, kernel_function="rbf_ard",
                      noise_function=lambda xx, h: np.where(xx[:, 0] > 2.0, -1.5, 1e-2))
    with pytest.raises(fvgp_amd.NonPositiveDefiniteError):
        bad.update_gp_data(x[:5] + 3.0, y[:5])


def test_cholinv_mode_variance_fast_path():
    """linalg_mode='CholInv' (gp.py:226-228, gp_kv.py:429-432, gp_posterior.py:238-244): cached KV^-1,
    variance_only posterior without forming S; everything else as in 'Chol'."""
    import fvgp_amd
    fx = load_golden("G3_matern52_n512_d3.npz")
    gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                     kernel_function="matern52_ard", linalg_mode="CholInv")
    inv = np.linalg.inv(orc.addKV(orc.matern52_ard(fx["x"], fx["x"], fx["theta"]), fx["noise_variances"]))
    got = gp._KVinv[:512, :512].cpu().numpy()
    assert np.max(np.abs(got - inv)) <= 1e-9 * np.max(np.abs(inv))
    pv = gp.posterior_covariance(fx["x_pred"], variance_only=True)
    assert pv["S"] is None and pv["S_flat"] is None
    assert np.max(np.abs(pv["v(x)"] - fx["pv"])) <= 1e-10
    pvn = gp.posterior_covariance(fx["x_pred"], variance_only=True, add_noise=True)
    assert np.max(np.abs(pvn["v(x)"] - fx["pv_noise"])) <= 1e-10
    full = gp.posterior_covariance(fx["x_pred"])
    assert np.max(np.abs(full["S"] - fx["pS"])) <= 1e-10
    np.testing.assert_allclose(gp.log_likelihood(fx["thetas"][1]), fx["logliks"][1], rtol=1e-10)
    with pytest.raises(NotImplementedError):
        fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"], linalg_mode="sparseCG")


def test_tiny_and_many_column_problems():
    """Edge sizes: fewer points than one 128 tile, a single point, and more y columns than the vector-solve path takes."""
    import fvgp_amd
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 127, 129):
        x = rng.random((n, 2)); y = np.sin(x.sum(axis=1)) + 0.01 * rng.standard_normal(n)
        nv = np.full(n, 0.05); th = np.array([0.8, 0.4, 0.6])
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="matern32_ard")
        ref = orc.OracleGP(x, y, th, nv, kernel="matern32_ard")
        np.testing.assert_allclose(gp.log_likelihood(), ref.log_likelihood(), rtol=1e-11)
        np.testing.assert_allclose(gp.log_likelihood(th * 1.1), ref.log_likelihood(th * 1.1), rtol=1e-11)
        np.testing.assert_allclose(gp.neg_log_likelihood_gradient(th), ref.neg_log_likelihood_gradient(th), rtol=1e-8, atol=1e-10)
        xp = rng.random((3, 2))
        np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], ref.posterior_mean(xp)["m(x)"], rtol=1e-9, atol=1e-12)
        assert np.max(np.abs(gp.posterior_covariance(xp)["S"] - ref.posterior_covariance(xp)["S"])) < 1e-11
    n, c = 300, 11
    x = rng.random((n, 3)); Y = np.stack([np.sin((k + 1) * x.sum(axis=1)) for k in range(c)], axis=1)
    nv = np.full(n, 0.02); th = np.array([1.0, 0.4, 0.5, 0.6])
    gp = fvgp_amd.GP(x, Y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    ref = orc.OracleGP(x, Y, th, nv, kernel="rbf_ard")
    np.testing.assert_allclose(gp.log_likelihood(), ref.log_likelihood(), rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(th * 0.9), ref.log_likelihood(th * 0.9), rtol=1e-10)
    assert np.max(np.abs(gp.KVinvY - ref.KVinvY)) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    xp = rng.random((4, 3))
    np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    assert gp.posterior_covariance(xp)["v(x)"].shape == (4, c)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(th, component=7), ref.neg_log_likelihood_gradient(th, component=7), rtol=1e-8)


def test_train_methods_improve_the_likelihood():
    """GP.train (gp.py:781) with the Dask-free methods; every objective call is a device evaluation."""
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=np.array([0.5, 0.8, 0.8, 0.8]),
                     noise_variances=fx["noise_variances"], kernel_function="rbf_ard")
    start = gp.log_likelihood()
    bounds = np.array([[0.05, 5.0]] + [[0.05, 2.0]] * 3)
    hps = gp.train(hyperparameter_bounds=bounds, method="local", max_iter=40)
    assert gp.log_likelihood() > start + 1.0 and np.allclose(hps, gp.hyperparameters)
    gp.set_hyperparameters(np.array([0.5, 0.8, 0.8, 0.8]))
    gp.train(hyperparameter_bounds=bounds, method="mcmc", max_iter=150, seed=1)
    assert gp.log_likelihood() > start
    assert "median(x)" in gp.mcmc_info
    gp.set_hyperparameters(np.array([0.5, 0.8, 0.8, 0.8]))
    gp.train(hyperparameter_bounds=bounds, method="global", max_iter=2, pop_size=4, seed=1)
    assert gp.log_likelihood() > start


def test_train_walks_the_reference_mcmc_chain_and_adam_history():
    """GP.train(method='mcmc' / 'adam') on the device likelihood against the reference's own seeded traces (G11): the
    chain is the same chain (positions from the same legacy random stream, the same accept decisions), not merely a
    better likelihood."""
    import fvgp_amd
    fx = load_golden("G11_training_traces_m52_n200_d2.npz")
    gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                     kernel_function="matern52_ard")
    got = gp.train(hyperparameter_bounds=fx["bounds"], init_hyperparameters=fx["theta"], method="mcmc",
                   max_iter=int(fx["mcmc_max_iter"]), seed=int(fx["seed"]))
    assert gp.mcmc_info["x"].shape == fx["mcmc_x"].shape
    np.testing.assert_allclose(gp.mcmc_info["x"], fx["mcmc_x"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(gp.mcmc_info["f(x)"], fx["mcmc_f"], rtol=1e-10)
    np.testing.assert_allclose(got, fx["mcmc_result"], rtol=1e-12)
    np.testing.assert_allclose(gp.hyperparameters, fx["mcmc_result"], rtol=1e-12)
    gp.set_hyperparameters(fx["theta"])
    got = gp.train(hyperparameter_bounds=fx["bounds"], init_hyperparameters=fx["theta"], method="adam",
                   max_iter=int(fx["adam_max_iter"]), accept_only_if_improved=False)
    np.testing.assert_allclose(np.asarray(gp.adam_history["theta"]), fx["adam_theta"], rtol=1e-7)
    np.testing.assert_allclose(np.asarray(gp.adam_history["nlml"]), fx["adam_nlml"], rtol=1e-9)
    np.testing.assert_allclose(got, fx["adam_result"], rtol=1e-7)
    # the guard: a training result that lowers the likelihood is rejected and the incumbent kept (gp.py:1086-1168)
    gp.set_hyperparameters(fx["adam_result"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        kept = gp.train(hyperparameter_bounds=fx["bounds"], init_hyperparameters=fx["theta"], method="adam", max_iter=1)
    assert any("rejected" in str(i.message) for i in w)
    np.testing.assert_array_equal(kept, fx["adam_result"])


def test_named_kernel_is_a_reference_style_callable():
    from fvgp_amd import kernels
    fx = load_golden("G8_iso_and_units.npz")
    for nm in ("rbf_iso", "matern32_iso", "matern52_iso"):
        K = kernels.NATIVE[nm](fx["x"], fx["x"], fx["theta"])
        assert np.max(np.abs(K[:8, :8] - fx[nm + "_K_corner"])) <= 8e-16 * fx["theta"][0]
        np.testing.assert_allclose(np.linalg.norm(K), float(fx[nm + "_K_fro"]), rtol=1e-13)


def test_finite_difference_derivatives_match_the_reference():
    """posterior_mean_grad, posterior_covariance_grad, Hessian, gradient self-test against the reference's own
    outputs.  The reference differences kernel matrices with a step of 1e-8, so its values carry ~1e-6 of
    rounding noise themselves; the bars below are that noise level, not the fp64 bars of the path."""
    import fvgp_amd
    fx = load_golden("G9_derivatives_rbf_n256_d2.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                         kernel_function="rbf_ard")
    xp = fx["x_pred"]
    sm = np.max(np.abs(fx["dm_all"]))
    r = gp.posterior_mean_grad(xp)
    assert r["direction"] == "ALL" and r["dm/dx"].shape == fx["dm_all"].shape
    np.testing.assert_allclose(r["dm/dx"], fx["dm_all"], rtol=0, atol=2e-5 * sm)
    r = gp.posterior_mean_grad(xp, direction=1)
    assert r["direction"] == 1
    np.testing.assert_allclose(r["dm/dx"], fx["dm_dir1"], rtol=0, atol=2e-5 * sm)
    np.testing.assert_allclose(gp.posterior_mean_grad(xp, hyperparameters=fx["theta2"])["dm/dx"], fx["dm_theta2"], rtol=0, atol=2e-5 * sm)
    sv = np.max(np.abs(fx["dS_dir0"]))
    np.testing.assert_allclose(gp.posterior_covariance_grad(xp)["dv/dx"], fx["dv_all"], rtol=0, atol=5e-5 * sv)
    r = gp.posterior_covariance_grad(xp, direction=0)
    np.testing.assert_allclose(r["dv/dx"], fx["dv_dir0"], rtol=0, atol=5e-5 * sv)
    np.testing.assert_allclose(r["dS/dx"], fx["dS_dir0"], rtol=0, atol=5e-5 * sv)
    hs = np.max(np.abs(fx["hessian"]))
    np.testing.assert_allclose(gp.neg_log_likelihood_hessian(fx["theta"]), fx["hessian"], rtol=0, atol=1e-4 * hs)
    fd, an = gp.test_log_likelihood_gradient(fx["theta"])
    np.testing.assert_allclose(an, fx["an_grad"], rtol=1e-8)
    np.testing.assert_allclose(fd, fx["fd_grad"], rtol=0, atol=1e-4 * np.max(np.abs(fx["fd_grad"])))
    # multi-task shapes (x_out reshapes)
    fm = load_golden("G9m_derivatives_fvgp_4x64.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gm = fvgp_amd.fvGP(fm["fvgp_x"], fm["fvgp_y"], init_hyperparameters=fm["theta"], noise_variances=fm["fvgp_noise"])
    xp5, xo = fm["x_pred"], fm["x_out"]
    s5 = np.max(np.abs(fm["dm_all"]))
    np.testing.assert_allclose(gm.posterior_mean_grad(xp5, x_out=xo)["dm/dx"], fm["dm_all"], rtol=0, atol=2e-5 * s5)
    np.testing.assert_allclose(gm.posterior_mean_grad(xp5, x_out=xo, direction=0)["dm/dx"], fm["dm_dir0"], rtol=0, atol=2e-5 * s5)
    s6 = np.max(np.abs(fm["dS_dir1"]))
    np.testing.assert_allclose(gm.posterior_covariance_grad(xp5, x_out=xo)["dv/dx"], fm["dv_all"], rtol=0, atol=5e-5 * s6)
    r = gm.posterior_covariance_grad(xp5, x_out=xo, direction=1)
    np.testing.assert_allclose(r["dv/dx"], fm["dv_dir1"], rtol=0, atol=5e-5 * s6)
    np.testing.assert_allclose(r["dS/dx"], fm["dS_dir1"], rtol=0, atol=5e-5 * s6)


def test_validation_scores_and_information_measures_match_the_reference():
    """The facade's scores (fvgp_amd/gp_validation.py) on the device posterior against the reference's values."""
    import fvgp_amd
    fx = load_golden("G10_scores_rbf_n400_d2.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                         kernel_function="rbf_ard")
    xt, yt = fx["x_test"], fx["y_test"]
    got = {"rmse": gp.rmse(xt, yt), "nrmse": gp.nrmse(xt, yt), "mae": gp.mae(xt, yt), "mape": gp.mape(xt, yt), "r2": gp.r2(xt, yt),
           "nlpd": gp.nlpd(xt, yt), "msll": gp.msll(xt, yt), "crps_mean": gp.crps(xt, yt)[0], "crps_std": gp.crps(xt, yt)[1],
           "picp": gp.picp(xt, yt), "mpiw": gp.mpiw(xt), "interval_score": gp.interval_score(xt, yt)}
    for k, v in got.items():
        np.testing.assert_allclose(v, fx["score_" + k], rtol=1e-7, err_msg=k)
    cc = gp.coverage_curve(xt, yt)
    np.testing.assert_allclose(cc["target_coverage"], fx["coverage_target"])
    np.testing.assert_allclose(cc["measured_coverage"], fx["coverage_measured"], atol=1e-12)
    xq, cm, cq = fx["x_q"], fx["comp_mean"], fx["comp_cov"]
    np.testing.assert_allclose(gp.gp_kl_div(xq, cm, cq)["kl-div"], fx["info_kl_div"], rtol=1e-6)
    np.testing.assert_allclose(gp.gp_relative_information_entropy(xq)["RIE"], fx["info_rie"], rtol=1e-6)
    np.testing.assert_allclose(gp.gp_relative_information_entropy_set(xq)["RIE"], fx["info_rie_set"], rtol=1e-6)
    pp = gp.posterior_probability(xq, cm, cq)
    np.testing.assert_allclose(pp["mu"], fx["info_pp_mu"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(pp["covariance"], fx["info_pp_cov"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(pp["probability"], fx["info_pp_prob"], rtol=1e-6)
    assert np.array_equal(fvgp_amd.GP.make_2d_x_pred([0, 1], [2, 3], 4, 3), fx["grid2d"])
    assert np.array_equal(fvgp_amd.GP.make_1d_x_pred([0, 2], 5), fx["grid1d"])


@pytest.mark.parametrize("P", [1, 2, 4, 5, 8])
def test_posterior_few_points_agree_with_the_oracle(P):
    """A handful of prediction points (acquisition-function optimisers ask for one at a time) runs the same block sweep as many."""
    import fvgp_amd
    fx = load_golden("G2_rbf_n512_d3.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                         kernel_function="rbf_ard")
    o = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel="rbf_ard")
    xp = fx["x_pred"][:P]
    got, want = gp.posterior_covariance(xp), o.posterior_covariance(xp)
    np.testing.assert_allclose(got["S"], want["S"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(got["v(x)"], want["v(x)"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], o.posterior_mean(xp)["m(x)"], rtol=1e-9)
    gotn = gp.posterior_covariance(xp, add_noise=True)
    np.testing.assert_allclose(gotn["S"], o.posterior_covariance(xp, add_noise=True)["S"], rtol=0, atol=1e-10)


def test_posterior_at_one_point_with_two_columns_of_y():
    """One prediction point takes the column + forward-sweep path; with y of shape (N, 2) the mean has one entry per column
    (gp_posterior.py:158) -- against the reference's own output for the first point of fixture G6."""
    fx = load_golden("G6_rbf_2col_n300_d3.npz")
    gp = _make(fx, {})
    xp = fx["x_pred"][:1]
    np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], fx["pm"][:1], rtol=1e-8, atol=1e-10)
    pc = gp.posterior_covariance(xp)
    assert np.max(np.abs(pc["v(x)"] - fx["pv"][:1])) <= 1e-10 * fx["theta"][0] + 1e-12
    assert np.max(np.abs(pc["S"] - fx["pS"][:1, :1])) <= 1e-10 * fx["theta"][0] + 1e-12
    pv = gp.posterior_covariance(xp, variance_only=True)
    assert np.max(np.abs(pv["v(x)"] - fx["pv"][:1])) <= 1e-10 * fx["theta"][0] + 1e-12


@pytest.mark.parametrize("n,m", [(2500, 4), (2048, 1)])
def test_append_matches_a_fresh_gp_on_a_long_factor(n, m):
    """update_gp_data(append=True) (gp_kv.py:462-476: bordering, v = L^-1 b by fvgp_hip_trsm_lower with few columns against at least
    2048 rows -> the transposed block sweep) against a fresh GP on the concatenated data."""
    import fvgp_amd
    rng = np.random.default_rng(n + m)
    x = rng.random((n + m, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n + m)
    th = np.array([1.1, 0.3, 0.35, 0.4]); nv = np.full(n + m, 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
        a.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
        b = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    xp = rng.random((7, 3))
    np.testing.assert_allclose(a.log_likelihood(), b.log_likelihood(), rtol=1e-10)
    assert np.max(np.abs(a.posterior_covariance(xp)["S"] - b.posterior_covariance(xp)["S"])) < 1e-10 * th[0]
    np.testing.assert_allclose(a.posterior_mean(xp)["m(x)"], b.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)


def test_facade_at_a_size_without_padding_rows():
    """N = 1024 (a multiple of 128: padded_dim(N) leaves no row for the appended (y-m)^T): the facade's square buffers get one more
    block row (fvgp_hip_loglik_dim) and the whole method surface -- likelihood at the state's and at another theta, gradient,
    posterior, CholInv variance, append, pickle -- answers as the oracle does."""
    import pickle
    import fvgp_amd
    from oracle import fvgp_oracle as orc
    n = 1024
    rng = np.random.default_rng(77)
    x = rng.random((n + 3, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n + 3)
    th = np.array([1.2, 0.3, 0.35, 0.4]); nv = np.full(n + 3, 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
    assert gp._L.shape == (n + 128, n + 128)
    ref = orc.OracleGP(x[:n], y[:n], th, nv[:n], kernel="rbf_ard")
    np.testing.assert_allclose(gp.log_likelihood(), ref.log_likelihood(), rtol=1e-10)
    t2 = th * np.array([1.3, 0.8, 1.1, 0.9])
    np.testing.assert_allclose(gp.log_likelihood(t2), ref.log_likelihood(t2), rtol=1e-10)
    assert np.max(np.abs(gp.KVinvY - ref.KVinvY)) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    g, g_ref = gp.neg_log_likelihood_gradient(th), ref.neg_log_likelihood_gradient(th)
    np.testing.assert_allclose(g, g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))
    xp = rng.random((9, 3))
    pc, rc = gp.posterior_covariance(xp), ref.posterior_covariance(xp)
    assert np.max(np.abs(pc["S"] - rc["S"])) <= 1e-10 * th[0]
    np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    gp2 = pickle.loads(pickle.dumps(gp))
    np.testing.assert_allclose(gp2.log_likelihood(), ref.log_likelihood(), rtol=1e-10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
    ref2 = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    np.testing.assert_allclose(gp.log_likelihood(), ref2.log_likelihood(), rtol=1e-10)
    np.testing.assert_allclose(gp.log_likelihood(t2), ref2.log_likelihood(t2), rtol=1e-10)


def test_repeated_appends_stay_in_the_factor_buffer_until_it_is_full():
    """update_gp_data(append=True) step after step, as an autonomous experiment does (gp_kv.py:462-476): while the new rows fit into
    the padding rows of the factor's buffer the bordered factor is written in place (same buffer object, no N x N copy), across a
    128-row block boundary too; the step that no longer fits moves to a larger buffer.  After every step: a fresh GP's answers."""
    import fvgp_amd
    rng = np.random.default_rng(123)
    n0, steps = 2400, [30, 70, 50, 10, 40]              # 2400 (buffer of 2432 rows) -> 2430 -> 2500 (new buffer, with headroom) -> 2550 -> 2560 -> 2600
    ntot = n0 + sum(steps)
    x = rng.random((ntot, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(ntot)
    th = np.array([1.1, 0.3, 0.35, 0.4]); nv = np.full(ntot, 0.01)
    xp = rng.random((6, 3))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = fvgp_amd.GP(x[:n0], y[:n0], init_hyperparameters=th, noise_variances=nv[:n0], kernel_function="rbf_ard")
        a.log_likelihood(th * 1.01)                      # (allocates the scratch pair an append in place keeps)
        n = n0
        same_buffer = []
        for m in steps:
            buf = a._L.data_ptr()
            a.update_gp_data(x[n:n + m], y[n:n + m], noise_variances_new=nv[n:n + m], append=True)
            n += m
            same_buffer.append(a._L.data_ptr() == buf)
            b = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
            np.testing.assert_allclose(a.log_likelihood(), b.log_likelihood(), rtol=1e-10)
            np.testing.assert_allclose(a.log_likelihood(th * 1.02), b.log_likelihood(th * 1.02), rtol=1e-10)
            assert np.max(np.abs(a.posterior_covariance(xp)["S"] - b.posterior_covariance(xp)["S"])) < 1e-10 * th[0]
            np.testing.assert_allclose(a.posterior_mean(xp)["m(x)"], b.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(a.neg_log_likelihood_gradient(th), b.neg_log_likelihood_gradient(th), rtol=1e-7, atol=1e-8)
            del b
    assert same_buffer == [True, False, True, True, True]


@pytest.mark.parametrize("chunk,budget,groups", [(256, None, 1), (256, 3 * 256 * 1536 * 8, 3), (128, 8 << 20, None)])
def test_posterior_covariance_in_chunks_of_prediction_points(chunk, budget, groups):
    """Many prediction points with bounded device memory (gp_posterior.py:120-136,229-288 hold k, L^-1 k and S in one piece): the
    points go through the device in chunks, the off-diagonal blocks of S are k(x_i, x_j) - V_i^T V_j from the chunks' L^-1 k.  Small
    chunk sizes stand in for the 4096 of production: S on the device (one group), S on the host with the chunks walked in two groups
    (a group's L^-1 k recomputed), and a budget so small that every chunk is a group -- against the oracle's one-piece result."""
    import fvgp_amd
    rng = np.random.default_rng(77)
    n, P = 1500, 700
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    th = np.array([1.2, 0.3, 0.35, 0.4]); nv = np.full(n, 0.01)
    xp = rng.random((P, 3))
    args = {"posterior_chunk": chunk}
    if budget is not None:
        args["posterior_scratch_bytes"] = budget
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard", args=args)
        one = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    o = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    got, want = gp.posterior_covariance(xp), o.posterior_covariance(xp)
    if groups is not None:
        assert gp._posterior_groups == groups
    else:
        assert gp._posterior_groups >= 3
    assert got["S"].shape == (P, P) and np.array_equal(got["S"], got["S"].T)
    assert np.max(np.abs(got["S"] - want["S"])) <= 1e-10 * th[0]
    assert np.max(np.abs(got["v(x)"] - want["v(x)"])) <= 1e-10 * th[0]
    # the one-piece device call (P below the default chunk) and the chunked one agree far inside the tolerance
    assert np.max(np.abs(got["S"] - one.posterior_covariance(xp)["S"])) <= 1e-12 * th[0]
    gn = gp.posterior_covariance(xp, add_noise=True, variance_only=True)
    np.testing.assert_allclose(gn["v(x)"], o.posterior_covariance(xp, add_noise=True)["v(x)"], rtol=0, atol=1e-10)
    if chunk == 128:
        # fewer data points than a chunk has prediction points (the scratch's two views, padded N x padded P_i and its transpose,
        # then differ in their leading dimension: tools/fuzz_posterior.py found the case), and a last chunk of a few points
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            small = fvgp_amd.GP(x[:300], y[:300], init_hyperparameters=th, noise_variances=nv[:300], kernel_function="rbf_ard",
                                args={"posterior_chunk": 512})
        o2 = orc.OracleGP(x[:300], y[:300], th, nv[:300], kernel="rbf_ard")
        xq = rng.random((530, 3))
        assert np.max(np.abs(small.posterior_covariance(xq)["S"] - o2.posterior_covariance(xq)["S"])) <= 1e-10 * th[0]
        np.testing.assert_allclose(small.posterior_mean(xq)["m(x)"], o2.posterior_mean(xq)["m(x)"], rtol=1e-8, atol=1e-10)


def test_failed_append_leaves_the_object_as_it_was():
    """update_gp_data(append=True) is all or nothing (cholesky_update_rank_n raises when the Schur complement is not positive definite,
    gp_lin_alg.py:1310-1477): after a failure in the middle -- here the factorisation of the Schur complement reports a non-positive
    pivot -- data, factor (the padding rows it had started to write included) and every cached result are the ones from before."""
    import fvgp_amd
    from fvgp_amd.gp_lin_alg import NonPositiveDefiniteError
    rng = np.random.default_rng(5)
    n, m = 900, 20
    x = rng.random((n + m, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n + m)
    th = np.array([1.1, 0.3, 0.35, 0.4]); nv = np.full(n + m, 0.01)
    xp = rng.random((5, 3))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(x[:n], y[:n], init_hyperparameters=th, noise_variances=nv[:n], kernel_function="rbf_ard")
    before = (gp.log_likelihood(), gp.posterior_covariance(xp)["S"].copy(), gp.KVinvY.copy(), gp._L.data_ptr(), gp._L.clone())
    H = gp._H
    real_potrf, real_potrs = H.potrf, H.potrs
    try:
        H.potrf = lambda S, mm: 3                                    # the Schur complement "is not positive definite"
        with pytest.raises(NonPositiveDefiniteError):
            gp.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
        H.potrf = real_potrf

        def boom(*a, **k):
            raise _lib_error("injected failure after the factor's rows were written")
        from fvgp_amd._lib import HipExtensionError as _lib_error
        H.potrs = boom                                               # fails AFTER the in-place rows have been written
        with pytest.raises(_lib_error):
            gp.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
    finally:
        H.potrf, H.potrs = real_potrf, real_potrs
    assert gp.point_number == n and len(gp.x_data) == n and len(gp.noise_variances) == n and gp._L.data_ptr() == before[3]
    assert torch_equal_lower(gp._L, before[4], gp._L.shape[0])
    assert gp.log_likelihood() == before[0] and np.array_equal(gp.KVinvY, before[2])
    assert np.max(np.abs(gp.posterior_covariance(xp)["S"] - before[1])) <= 1e-13
    # ... and the append still works afterwards
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp.update_gp_data(x[n:], y[n:], noise_variances_new=nv[n:], append=True)
        fresh = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function="rbf_ard")
    np.testing.assert_allclose(gp.log_likelihood(), fresh.log_likelihood(), rtol=1e-10)


def torch_equal_lower(a, b, n):
    import torch
    return bool(torch.equal(torch.tril(a[:n, :n]), torch.tril(b[:n, :n])))
