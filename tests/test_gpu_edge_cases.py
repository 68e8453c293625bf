"""Edge cases the reference's own suite holds for the path and its callers (tests/test_fvgp.py, cited per test),
replayed on fvgp_amd.GP / fvGP: same constructor arguments, same calls, same messages.  Values are checked against the
oracle where the reference's test only checks shapes."""
import warnings

import numpy as np
import pytest

from oracle import fvgp_oracle as orc

pytestmark = pytest.mark.gpu


def _tiny_gp(**kw):
    """tests/test_fvgp.py `_tiny_gp`: 12 points in 2-d, default kernel, fixed noise."""
    import fvgp_amd
    rng = np.random.default_rng(11)
    xx = rng.random((12, 2))
    yy = np.sin(np.linalg.norm(xx, axis=1))
    kw.setdefault("noise_variances", np.full(12, 0.01))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fvgp_amd.GP(xx, yy, np.array([1., 1., 1.]), **kw)


def _tiny_fvgp():
    import fvgp_amd
    rng = np.random.default_rng(12)
    x2 = rng.random((12, 2))
    r = np.linalg.norm(x2, axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fvgp_amd.fvGP(x2, np.column_stack([np.sin(r), np.cos(r)]), np.array([1., 1., 1., 1.]),
                             noise_variances=np.full((12, 2), 0.01))


def test_data_update_rejects_bad_noise_combinations():
    """tests/test_fvgp.py:3578-3599 (GPdata.update, gp_data.py:84-89)"""
    import fvgp_amd
    rng = np.random.default_rng(1)
    xx = rng.random((10, 2))
    yy = np.sin(np.linalg.norm(xx, axis=1))
    v = np.ones(10) * 0.01
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with_noise = fvgp_amd.GP(xx, yy, np.array([1., 1., 1.]), noise_variances=v)
        without_noise = fvgp_amd.GP(xx, yy, np.array([1., 1., 1.]))
    with pytest.raises(Exception, match="Please provide noise_variances"):
        with_noise.update_gp_data(xx[:2] + 0.5, yy[:2], None, append=True)
    with pytest.raises(Exception, match="did not initialize noise"):
        without_noise.update_gp_data(xx[:2] + 0.5, yy[:2], v[:2], append=True)
    # nothing was appended by the rejected calls
    assert with_noise.point_number == 10 and without_noise.point_number == 10
    # overwrite + rank-n update is contradictory: warned, then recomputed (gp.py:733-737)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with_noise.update_gp_data(xx[:6], yy[:6], v[:6], append=False, rank_n_update=True)
    assert any("rank_n_update=True" in str(w.message) for w in caught)
    assert with_noise.point_number == 6
    ref = orc.OracleGP(xx[:6], yy[:6], np.array([1., 1., 1.]), noise_variances=v[:6])
    np.testing.assert_allclose(with_noise.log_likelihood(), ref.log_likelihood(), rtol=1e-10)


def test_likelihood_branches():
    """tests/test_fvgp.py:3626-3660: ambiguous noise, the three-argument noise function"""
    import fvgp_amd
    rng = np.random.default_rng(2)
    xx = rng.random((12, 2))
    yy = np.sin(np.linalg.norm(xx, axis=1))
    with pytest.raises(Exception, match="Decide which one to use"):
        fvgp_amd.GP(xx, yy, np.array([1., 1., 1.]), noise_variances=np.full(12, 0.01),
                    noise_function=lambda x, hps: np.full(len(x), 0.01))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp3 = fvgp_amd.GP(xx, yy, np.array([1., 1., 1.]),
                          noise_function=lambda x, hps, args: np.full(len(x), args["level"]), args={"level": 0.02})
    assert np.allclose(gp3.V, 0.02)
    ref = orc.OracleGP(xx, yy, np.array([1., 1., 1.]), noise_variances=np.full(12, 0.02))
    np.testing.assert_allclose(gp3.log_likelihood(), ref.log_likelihood(), rtol=1e-10)


def test_fvgp_rejects_single_task_y_data():
    """tests/test_fvgp.py:3663-3671"""
    import fvgp_amd
    x2 = np.random.default_rng(3).random((10, 2))
    with pytest.raises(ValueError, match="output number is 1"):
        fvgp_amd.fvGP(x2, np.sin(np.linalg.norm(x2, axis=1)), np.array([1., 1., 1., 1.]))


def test_fvgp_update_data_formats_and_nan_skipping():
    """tests/test_fvgp.py:3674-3695: NaN entries never enter the index set; the append path keeps the (V, Di) view of
    the data and validates formats"""
    import fvgp_amd
    rng = np.random.default_rng(4)
    x2 = rng.random((12, 2))
    r = np.linalg.norm(x2, axis=1)
    y2 = np.column_stack([np.sin(r), np.cos(r)])
    y2[0, 1] = np.nan
    v2 = np.full(y2.shape, 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.fvGP(x2, y2, np.array([1., 1., 1., 1.]), noise_variances=v2)
    assert len(gp.x_data) == y2.size - 1
    x_add = rng.random((2, 2))
    ra = np.linalg.norm(x_add, axis=1)
    y_add = np.column_stack([np.sin(ra), np.cos(ra)])
    gp.update_gp_data(x_add, y_add, np.full(y_add.shape, 0.01), append=True)
    assert len(gp.fvgp_x_data) == 14 and gp.fvgp_y_data.shape == (14, 2) and gp.fvgp_noise_variances.shape == (14, 2)
    assert len(gp.x_data) == y2.size - 1 + 4
    with pytest.raises((Exception, AssertionError), match="format in x_new"):
        gp.update_gp_data("not an array", y_add, np.full(y_add.shape, 0.01), append=True)
    # the appended model equals one built on all the data at once (task-major order differs, the likelihood does not)
    xa, ya, va = np.vstack([x2, x_add]), np.vstack([y2, y_add]), np.full((14, 2), 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        whole = fvgp_amd.fvGP(xa, ya, np.array([1., 1., 1., 1.]), noise_variances=va)
    np.testing.assert_allclose(gp.log_likelihood(), whole.log_likelihood(), rtol=1e-10)
    xp = rng.random((5, 2))
    np.testing.assert_allclose(gp.posterior_mean(xp)["m(x)"], whole.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)


def test_marginal_likelihood_reports_linalg_failures():
    """tests/test_fvgp.py:3738-3753: a failure inside the evaluation is reported with its hyperparameters"""
    gp = _tiny_gp()

    def boom(*a, **k):
        raise RuntimeError("factorization exploded")
    original = gp._H.loglik
    gp._H.loglik = boom
    try:
        with pytest.raises(Exception) as e:
            gp.log_likelihood(hyperparameters=gp.hyperparameters * 1.1)
        assert "Linear algebra failed" in str(e.value) and "factorization exploded" in str(e.value)
    finally:
        gp._H.loglik = original
    assert np.isfinite(gp.log_likelihood(hyperparameters=gp.hyperparameters * 1.1))


def test_training_rejects_bad_starting_points_and_methods():
    """tests/test_fvgp.py:3778-3805 (GPtraining.train, gp_training.py:54-55,194-196)"""
    from fvgp_amd import gp_training
    gp = _tiny_gp()
    bounds = np.array([[0.01, 10.], [0.01, 10.], [0.01, 10.]])
    with pytest.raises(Exception, match="outside of optimization bounds"):
        gp_training.train(gp, bounds, np.array([100., 100., 100.]), method="local")
    with pytest.raises(ValueError, match="No optimization mode"):
        gp_training.train(gp, bounds, np.array([1., 1., 1.]), method=42)
    with pytest.raises(AssertionError, match="invalid hyperparameters"):
        gp_training.train(gp, bounds, np.array([1., 1., 1.]), method=lambda g: [1.0, 1.0, 1.0])


def test_train_argument_validation():
    """tests/test_fvgp.py:3995-4028 (GP.train, gp.py:1007-1053)"""
    gp = _tiny_gp()
    bounds = np.array([[0.01, 10.], [0.01, 10.], [0.01, 10.]])
    with pytest.raises(Exception, match="dask_client"):
        gp.train(hyperparameter_bounds=bounds, method="mcmc", asynchronous=True, max_iter=2)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        gp.train(hyperparameter_bounds=bounds, method="local", max_iter=1, init_hyperparameters=np.array([500., 500., 500.]))
    assert any("out of bounds" in str(w.message) for w in caught)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        gp.train(hyperparameter_bounds=bounds, method="mcmc", max_iter=2, objective_function=lambda hps: 0.0)
    assert any("user-defined objective_function is ignored" in str(w.message) for w in caught)
    with pytest.raises(Exception, match="gradient"):
        gp.train(hyperparameter_bounds=bounds, method="local", max_iter=1, objective_function=lambda hps: 0.0)
    # no bounds: the data-derived default box, with a warning (gp.py:1019-1023)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        hps = gp.train(method="mcmc", max_iter=5)
    assert any("Default hyperparameter_bounds" in str(w.message) for w in caught)
    box = gp._default_bounds()
    assert np.all(hps >= box[:, 0]) and np.all(hps <= box[:, 1])


def test_user_objective_callable_method_and_mcmc_prior():
    """gp.py:1038-1053 / gp_training.py:194 / gp_mcmc.py:203-209: a user objective with its gradient drives 'local'; a
    callable method gets the GP; a log prior of -inf vetoes a proposal before the likelihood is evaluated"""
    gp = _tiny_gp()
    bounds = np.array([[0.01, 10.], [0.01, 10.], [0.01, 10.]])
    target = np.array([2.0, 0.5, 3.0])
    hps = gp.train(hyperparameter_bounds=bounds, method="local", max_iter=200, tolerance=1e-12,
                   objective_function=lambda h: float(np.sum((h - target) ** 2)),
                   objective_function_gradient=lambda h: 2.0 * (h - target))
    np.testing.assert_allclose(hps, target, atol=1e-5)
    np.testing.assert_allclose(gp.hyperparameters, target, atol=1e-5)
    # a callable that returns a worse point is rolled back (gp.py:1086-1168) ...
    gp.set_hyperparameters(np.array([1., 1., 1.]))
    bad = np.array([9.9, 0.011, 9.9])
    assert gp.log_likelihood(bad) < gp.log_likelihood()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        kept = gp.train(hyperparameter_bounds=bounds, method=lambda g: bad.copy())
    assert any("rejected" in str(w.message) for w in caught)
    np.testing.assert_array_equal(kept, [1., 1., 1.])
    # ... unless told otherwise
    got = gp.train(hyperparameter_bounds=bounds, method=lambda g: bad.copy(), accept_only_if_improved=False)
    np.testing.assert_array_equal(got, bad)
    # prior: only the lower half of the box for hps[0]
    gp.set_hyperparameters(np.array([1., 1., 1.]))
    calls = []

    def prior(theta, box, args):
        ok = bool(np.all(theta >= box[:, 0]) and np.all(theta <= box[:, 1]) and theta[0] < 2.0)
        calls.append(ok)
        return 0.0 if ok else -np.inf
    gp.train(hyperparameter_bounds=bounds, method="mcmc", max_iter=80, mcmc_prior=prior, seed=3)
    assert np.all(gp.mcmc_info["x"][:, 0] < 2.0) and len(calls) >= 80


def test_default_hyperparameter_bounds_refuse_custom_functions():
    """tests/test_fvgp.py:3981-3992"""
    import fvgp_amd
    from fvgp_amd import kernels
    rng = np.random.default_rng(5)
    xx = rng.random((10, 2))
    yy = np.sin(np.linalg.norm(xx, axis=1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(xx, yy, np.array([1., 1.]),
                         kernel_function=lambda x1, x2, hps: hps[0] * np.exp(
                             -kernels.get_anisotropic_distance_matrix(x1, x2, np.array([hps[1], hps[1]])) ** 2))
    with pytest.raises(Exception, match="custom hyperparameter_bounds"):
        gp._default_bounds()
    with pytest.raises(Exception, match="custom hyperparameter_bounds"):
        gp.train(method="mcmc", max_iter=3)


def test_four_argument_kernel_and_three_argument_mean():
    """tests/test_fvgp.py:3867-3893: the args-taking signatures, and an append under a custom mean"""
    import fvgp_amd
    from fvgp_amd import kernels
    rng = np.random.default_rng(6)
    xx = rng.random((15, 2))
    yy = np.sin(np.linalg.norm(xx, axis=1))
    hps = np.array([1., 1., 1.])

    def kernel_with_args(x1, x2, hps, args):
        return args["amp"] * np.exp(-kernels.get_anisotropic_distance_matrix(x1, x2, hps[1:]) ** 2)

    def mean_with_args(x, hps, args):
        return np.full(len(x), args["offset"])

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(xx, yy, hps, kernel_function=kernel_with_args, prior_mean_function=mean_with_args,
                         prior_mean_function_grad=lambda x, hps: np.zeros((len(hps), len(x))),
                         args={"amp": 2.0, "offset": 0.5})
    assert np.allclose(gp.m, 0.5)
    assert np.allclose(np.diag(gp.K), 2.0)
    # the same model through scipy on the host
    from scipy.linalg import cho_factor, cho_solve
    V = np.full(15, (np.mean(np.abs(yy)) / 100.0) ** 2)
    KV = kernel_with_args(xx, xx, hps, {"amp": 2.0}) + np.diag(V)
    c = cho_factor(KV, lower=True)
    a = cho_solve(c, yy - 0.5)
    ll = -0.5 * ((yy - 0.5) @ a + 2.0 * np.sum(np.log(np.diag(c[0]))) + 15 * np.log(2.0 * np.pi))
    np.testing.assert_allclose(gp.log_likelihood(), ll, rtol=1e-10)
    x_add = rng.random((2, 2))
    gp.update_gp_data(x_add, np.sin(np.linalg.norm(x_add, axis=1)), append=True)
    assert len(gp.m) == 17 and gp.point_number == 17


def test_posterior_variance_tiling_for_multi_column_y():
    """tests/test_fvgp.py:4319-4324 (gp_posterior.py:279-281)"""
    import fvgp_amd
    rng = np.random.default_rng(7)
    xx = rng.random((12, 2))
    y1 = np.sin(np.linalg.norm(xx, axis=1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp = fvgp_amd.GP(xx, np.tile(y1[:, None], (1, 2)), np.array([1., 1., 1.]), noise_variances=np.full(12, 0.01))
        one = fvgp_amd.GP(xx, y1, np.array([1., 1., 1.]), noise_variances=np.full(12, 0.01))
    xp = rng.random((5, 2))
    result = gp.posterior_covariance(xp)
    assert result["v(x)"].shape == (5, 2)
    np.testing.assert_allclose(result["v(x)"][:, 0], one.posterior_covariance(xp)["v(x)"], rtol=1e-10, atol=1e-14)
    np.testing.assert_array_equal(result["v(x)"][:, 0], result["v(x)"][:, 1])


def test_posterior_warns_and_clips_negative_variances():
    """tests/test_fvgp.py:4327-4340 (gp_posterior.py:248-259): warn below -1e-4, clip to 0, write the clipped diagonal
    back into S"""
    gp = _tiny_gp()
    xp = np.random.default_rng(8).random((4, 2))
    real = gp._posterior_device

    def inflated(x_pred, hps, L, alpha, want_cov):
        mean, S = real(x_pred, hps, L, alpha, want_cov)
        return mean, (None if S is None else S - np.eye(len(S)) * 10.0)
    gp._posterior_device = inflated
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        result = gp.posterior_covariance(xp)
    assert any("Negative variances" in str(w.message) for w in caught)
    assert np.all(result["v(x)"] >= 0.0)
    assert np.all(np.diag(result["S"]) >= 0.0)


def test_multi_task_posterior_reshape_paths():
    """tests/test_fvgp.py:4304-4316: the x_out branches of the posterior mean, its gradient and the covariance gradient"""
    gp = _tiny_fvgp()
    xp = np.random.default_rng(9).random((4, 2))
    assert gp.posterior_mean(xp)["m(x)"].shape == (4, 2)
    assert gp.posterior_mean_grad(xp, direction=0)["dm/dx"].shape == (4, 2)
    assert gp.posterior_mean_grad(xp)["dm/dx"].shape == (4, 2, 2)
    grad = gp.posterior_covariance_grad(xp, direction=0)
    assert grad["dv/dx"].shape == (4, 2)
    assert grad["dS/dx"].shape == (4, 4, 2, 2)
    assert gp.posterior_covariance_grad(xp)["dv/dx"].shape == (4, 2, 2)
    cov = gp.posterior_covariance(xp)
    assert cov["v(x)"].shape == (4, 2) and cov["S"].shape == (4, 4, 2, 2)
    one_task = gp.posterior_mean(xp, x_out=np.array([1.]))["m(x)"]
    np.testing.assert_allclose(one_task[:, 0], gp.posterior_mean(xp)["m(x)"][:, 1], rtol=1e-12)
