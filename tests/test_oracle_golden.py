"""The oracle (oracle/fvgp_oracle.py) held to the vectors the reference itself produced
(tests/golden/*.npz, made by oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import fvgp_oracle as orc

CASES = [
    ("G1_rbf_n500_d1.npz", {}),
    ("G2_rbf_n512_d3.npz", {}),
    ("G3_matern52_n512_d3.npz", {}),
    ("G4_default_n256_d2.npz", {"default_noise": True}),
    ("G5_fvgp_4x64.npz", {"x_out": True}),
    ("G5n_fvgp_4x64_nan.npz", {"x_out": True}),
    ("G6_rbf_2col_n300_d3.npz", {}),
]


def _gp(fx, opt):
    nv = None if opt.get("default_noise") else fx["noise_variances"]
    x_out = fx["x_out"] if opt.get("x_out") else None
    return orc.OracleGP(fx["x"], fx["y"], fx["theta"], nv, kernel=str(fx["kernel"]), x_out=x_out)


@pytest.mark.parametrize("name,opt", CASES)
def test_state_and_likelihood(name, opt):
    fx = load_golden(name)
    gp = _gp(fx, opt)
    assert np.array_equal(gp.K[:8, :8], fx["K_corner"])
    assert np.array_equal(gp.K[-1, :], fx["K_row_last"])
    assert gp.K.trace() == fx["K_trace"]
    L = np.tril(gp.Chol_factor)
    np.testing.assert_allclose(np.diag(L), fx["L_diag"], rtol=1e-13)
    np.testing.assert_allclose(L[-1, :], fx["L_row_last"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(gp.logdet_KV, fx["logdet"], rtol=1e-14)
    np.testing.assert_allclose(gp.KVinvY, fx["KVinvY"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gp.log_likelihood(), fx["loglik"], rtol=1e-14)
    np.testing.assert_allclose(gp.log_likelihood(fx["theta"]), fx["loglik_theta"], rtol=1e-14)
    for t, ll in zip(fx["thetas"], fx["logliks"]):
        np.testing.assert_allclose(gp.log_likelihood(t), ll, rtol=1e-13)


@pytest.mark.parametrize("name,opt", CASES)
def test_gradient(name, opt):
    fx = load_golden(name)
    gp = _gp(fx, opt)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"]), fx["grad"], rtol=1e-9)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(), fx["grad_cached"], rtol=1e-9)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient_potri(fx["theta"]), fx["grad"], rtol=1e-8, atol=1e-9)
    if "grad_c1" in fx:
        np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"], component=1), fx["grad_c1"], rtol=1e-9)


@pytest.mark.parametrize("name,opt", CASES)
def test_posterior(name, opt):
    fx = load_golden(name)
    gp = _gp(fx, opt)
    xp = fx["x_pred"]
    pm = gp.posterior_mean(xp)
    np.testing.assert_allclose(pm["m(x)"], fx["pm"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(pm["m(x)_flat"], fx["pm_flat"], rtol=1e-11, atol=1e-12)
    assert np.array_equal(pm["x_pred"], fx["pm_xpred"])
    pc = gp.posterior_covariance(xp)
    pcn = gp.posterior_covariance(xp, add_noise=True)
    for key, tag in (("v(x)", "pv"), ("S", "pS"), ("S_flat", "pS_flat"), ("v_flat", "pv_flat")):
        assert np.asarray(pc[key]).shape == fx[tag].shape
        np.testing.assert_allclose(pc[key], fx[tag], rtol=0, atol=1e-11)
        np.testing.assert_allclose(pcn[key], fx[tag + "_noise"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(gp.posterior_mean(xp, hyperparameters=fx["thetas"][0])["m(x)"], fx["pm_theta1"],
                               rtol=1e-10, atol=1e-11)


def test_appendix_a_anchor():
    """the value the survey recorded from the imported reference."""
    fx = load_golden("G0_appendixA.npz")
    assert abs(float(fx["loglik"]) - 607.5932505420312) < 1e-9
    gp = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel="rbf_ard")
    np.testing.assert_allclose(gp.log_likelihood(fx["theta"]), 607.5932505420312, rtol=1e-13)
    # the survey's gradient went through the reference's central-FD kernel gradient (eps 1e-8,
    # gp_prior.py:438-447), good to ~1e-5; the fixture used the analytic dK through the same route
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"]), [3.43091903, -98.62223411], rtol=2e-5)
    np.testing.assert_allclose(gp.neg_log_likelihood_gradient(fx["theta"]), fx["grad"], rtol=1e-9)


def test_nonpd_and_linalg_units():
    fx = load_golden("G7_nonpd.npz")
    with pytest.raises(orc.NonPositiveDefiniteError) as ei:
        orc.calculate_Chol_factor(fx["M"])
    assert isinstance(ei.value, np.linalg.LinAlgError)
    assert f"{int(fx['info'])}-th leading minor" in str(fx["message"])
    L = np.tril(orc.calculate_Chol_factor(fx["Mok"]))
    np.testing.assert_allclose(L, fx["Lok"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(orc.calculate_Chol_solve(L, fx["rhs"]), fx["sol"], rtol=1e-12)
    np.testing.assert_allclose(orc.calculate_Chol_logdet(L), fx["logdet"], rtol=1e-14)
    assert orc.calculate_Chol_solve(L, fx["rhs"][:, 0]).shape == (96, 1)


def test_kernel_units_and_iso():
    fx = load_golden("G8_iso_and_units.npz")
    x1, x2, ls = fx["x1"], fx["x2"], fx["lengths"]
    assert np.array_equal(orc.get_distance_matrix(x1, x2), fx["dist_iso"])
    assert np.array_equal(orc.get_anisotropic_distance_matrix(x1, x2, ls), fx["dist_aniso"])
    d = fx["dist_aniso"]
    assert np.array_equal(orc.squared_exponential_kernel(d, 0.8), fx["sqexp"])
    assert np.array_equal(orc.matern_kernel_diff1(d, 0.8), fx["mat1"])
    assert np.array_equal(orc.matern_kernel_diff2(d, 0.8), fx["mat2"])
    np.testing.assert_allclose(orc.matern32_ard_grad(x1, x2, fx["theta3"]), fx["grad_m32"], rtol=0, atol=0)
    np.testing.assert_allclose(orc.matern52_ard_grad(x1, x2, fx["theta3"]), fx["grad_m52"], rtol=1e-15)
    for nm in ("rbf_iso", "matern32_iso", "matern52_iso"):
        gp = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel=nm)
        assert np.array_equal(gp.K[:8, :8], fx[nm + "_K_corner"])
        np.testing.assert_allclose(gp.log_likelihood(fx["theta"]), fx[nm + "_loglik"], rtol=1e-13)


def test_multitask_transform():
    fx = load_golden("G5n_fvgp_4x64_nan.npz")
    xt, yt, vt = orc.transform_index_set(fx["fvgp_x"], fx["fvgp_y"], fx["fvgp_noise"])
    assert np.array_equal(xt, fx["x"]) and np.array_equal(yt, fx["y"][:, 0] if fx["y"].ndim == 2 else fx["y"])
    assert np.array_equal(vt, fx["noise_variances"])
    assert len(xt) == 4 * 64 - 2
    cp = orc.cartesian_product(fx["x_pred"], fx["x_out"])
    assert np.array_equal(cp, fx["pm_xpred"])


def test_finite_difference_derivatives_match_the_reference():
    """posterior_mean_grad / posterior_covariance_grad / Hessian / gradient self-test (gp_posterior.py:184-226,
    290-331; gp_marginal_likelihood.py:312-364) -- the oracle repeats the reference's arithmetic exactly."""
    fx = load_golden("G9_derivatives_rbf_n256_d2.npz")
    o = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel="rbf_ard")
    xp = fx["x_pred"]
    np.testing.assert_allclose(o.posterior_mean_grad(xp)["dm/dx"], fx["dm_all"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o.posterior_mean_grad(xp, direction=1)["dm/dx"], fx["dm_dir1"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o.posterior_mean_grad(xp, hyperparameters=fx["theta2"])["dm/dx"], fx["dm_theta2"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o.posterior_covariance_grad(xp)["dv/dx"], fx["dv_all"], rtol=0, atol=1e-9)
    r = o.posterior_covariance_grad(xp, direction=0)
    np.testing.assert_allclose(r["dv/dx"], fx["dv_dir0"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["dS/dx"], fx["dS_dir0"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o.neg_log_likelihood_hessian(fx["theta"]), fx["hessian"], rtol=1e-9, atol=1e-6)
    fd, an = o.test_log_likelihood_gradient(fx["theta"])
    np.testing.assert_allclose(fd, fx["fd_grad"], rtol=1e-9)
    np.testing.assert_allclose(an, fx["an_grad"], rtol=1e-9)
    # multi-task shapes
    fm = load_golden("G9m_derivatives_fvgp_4x64.npz")
    xt, yt, vt = orc.transform_index_set(fm["fvgp_x"], fm["fvgp_y"], fm["fvgp_noise"])
    om = orc.OracleGP(xt, yt, fm["theta"], vt, kernel="matern32_ard", x_out=fm["x_out"])
    xp5, xo = fm["x_pred"], fm["x_out"]
    assert om.posterior_mean_grad(xp5, x_out=xo)["dm/dx"].shape == fm["dm_all"].shape == (6, 2, 4)
    np.testing.assert_allclose(om.posterior_mean_grad(xp5, x_out=xo)["dm/dx"], fm["dm_all"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(om.posterior_covariance_grad(xp5, x_out=xo, direction=1)["dS/dx"], fm["dS_dir1"], rtol=0, atol=1e-9)


def test_validation_scores_and_information_measures_match_the_reference():
    """RMSE .. interval score (fvgp/gp.py:1754-2071) and the P x P KL / RIE / posterior-probability measures
    (gp_posterior.py:408-552) restated on the oracle's posterior."""
    fx = load_golden("G10_scores_rbf_n400_d2.npz")
    o = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel="rbf_ard")
    sc = orc.validation_scores(o, fx["x_test"], fx["y_test"])
    for k, v in sc.items():
        np.testing.assert_allclose(v, fx["score_" + k], rtol=1e-10, err_msg=k)
    info = orc.information_measures(o, fx["x_q"], fx["comp_mean"], fx["comp_cov"])
    for k, v in info.items():
        np.testing.assert_allclose(v, fx["info_" + k], rtol=1e-8, atol=1e-12, err_msg=k)
