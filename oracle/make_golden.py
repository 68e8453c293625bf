"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (build container only).

Test infrastructure.  Imports lbl-camera/fvGP from /root/reference with stub modules for
the packages this image lacks (loguru, dask, distributed, hgdl, fvgp._version -- SURVEY
Appendix A), evaluates the hot path on seeded inputs, cross-checks oracle/fvgp_oracle.py
against it, and freezes inputs + expected outputs as small .npz fixtures.

Only the arrays travel to the GPU box; /root/reference does not.  Re-run with
    python oracle/make_golden.py
whenever the oracle grows a new function.  Fixtures are data, not source.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("FVGP_REFERENCE", "/root/reference")


def import_reference():
    def _mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    class _NoClient:
        def __init__(self, *a, **k):
            raise RuntimeError("no dask in this image")

    _mk("loguru", logger=_Logger())
    _mk("distributed", Client=_NoClient)
    d = _mk("dask")
    d.distributed = _mk("dask.distributed", Client=_NoClient, get_worker=lambda: None,
                        as_completed=None, performance_report=None)
    h = _mk("hgdl")
    h.hgdl = _mk("hgdl.hgdl", HGDL=object)
    sys.path.insert(0, REF)
    _mk("fvgp._version", __version__="0+reference")
    import fvgp  # noqa
    return fvgp


def synth(n, d, seed=20240501, ncol=1):
    """SURVEY 8d synthetic inputs."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(3.0 * np.sum(x, axis=1)) + 0.1 * rng.standard_normal(n)
    if ncol > 1:
        cols = [y]
        for c in range(1, ncol):
            cols.append(np.cos((2.0 + c) * np.sum(x, axis=1)) + 0.1 * rng.standard_normal(n))
        y = np.stack(cols, axis=1)
    return x, y


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    warnings.simplefilter("ignore")
    fvgp = import_reference()
    from fvgp import kernels as rk
    from fvgp import gp_bo, gp_lin_alg
    sys.path.insert(0, ROOT)
    from oracle import fvgp_oracle as orc
    os.makedirs(OUT, exist_ok=True)
    report = []

    def ref_rbf(x1, x2, h):
        return h[0] * rk.squared_exponential_kernel(rk.get_anisotropic_distance_matrix(x1, x2, h[1:]), 1.)

    def ref_rbf_iso(x1, x2, h):
        return h[0] * rk.squared_exponential_kernel(rk.get_distance_matrix(x1, x2), h[1])

    def ref_m32_iso(x1, x2, h):
        return h[0] * rk.matern_kernel_diff1(rk.get_distance_matrix(x1, x2), h[1])

    def ref_m52_iso(x1, x2, h):
        return h[0] * rk.matern_kernel_diff2(rk.get_distance_matrix(x1, x2), h[1])

    def check(tag, a, b, tol):
        r = rel(a, b)
        report.append((tag, r, tol))
        assert r <= tol, f"oracle != reference for {tag}: rel {r:.3e} > {tol}"

    def common(gp, o, x, y, nv, theta, thetas, P, seed, ref_kernel, name, grad=True, x_out=None):
        """Run reference GP `gp` and oracle `o` through the hot methods; return fixture dict."""
        fx = {"x": x, "y": y, "noise_variances": nv, "theta": theta}
        K = gp.K
        L = np.tril(gp.kv.Chol_factor)
        fx.update(K_corner=K[:8, :8].copy(), K_trace=np.trace(K), K_fro=np.linalg.norm(K),
                  K_row_last=K[-1, :].copy(),
                  L_diag=np.diag(L).copy(), L_corner=L[:8, :8].copy(), L_row_last=L[-1, :].copy(),
                  logdet=gp.kv.logdet_KV, KVinvY=gp.kv.KVinvY.copy(), m=gp.prior.m.copy(), V=gp.likelihood.V.copy(),
                  loglik=gp.log_likelihood(), loglik_theta=gp.log_likelihood(theta))
        check(name + ".K", o.K, K, 1e-14)
        check(name + ".L", np.tril(o.Chol_factor), L, 1e-12)
        check(name + ".logdet", o.logdet_KV, gp.kv.logdet_KV, 1e-13)
        check(name + ".KVinvY", o.KVinvY, gp.kv.KVinvY, 1e-10)
        check(name + ".loglik", o.log_likelihood(), fx["loglik"], 1e-13)
        fx["thetas"] = np.asarray(thetas)
        fx["logliks"] = np.array([gp.log_likelihood(t) for t in thetas])
        check(name + ".logliks", [o.log_likelihood(t) for t in thetas], fx["logliks"], 1e-12)
        if grad:
            fx["grad"] = gp.neg_log_likelihood_gradient(theta)
            fx["grad_cached"] = gp.neg_log_likelihood_gradient()
            check(name + ".grad", o.neg_log_likelihood_gradient(theta), fx["grad"], 1e-9)
            check(name + ".grad_potri", o.neg_log_likelihood_gradient_potri(theta), fx["grad"], 1e-8)
        if P:
            rng = np.random.default_rng(seed + 1)
            xp = rng.random((P, x.shape[1] if x_out is None else x.shape[1] - 1))
            fx["x_pred"] = xp
            kw = {} if x_out is None else {"x_out": x_out}
            pm = gp.posterior_mean(xp, **kw)
            pc = gp.posterior_covariance(xp, **kw)
            pcn = gp.posterior_covariance(xp, add_noise=True, **kw)
            om = o.posterior_mean(xp, **kw)
            oc = o.posterior_covariance(xp, **kw)
            ocn = o.posterior_covariance(xp, add_noise=True, **kw)
            for key, tag in (("m(x)", "pm"), ("m(x)_flat", "pm_flat"), ("x_pred", "pm_xpred")):
                fx[tag] = np.asarray(pm[key])
                check(name + "." + tag, om[key], pm[key], 1e-10)
            for key, tag in (("v(x)", "pv"), ("S", "pS"), ("S_flat", "pS_flat"), ("v_flat", "pv_flat")):
                fx[tag] = np.asarray(pc[key])
                fx[tag + "_noise"] = np.asarray(pcn[key])
                assert np.max(np.abs(np.asarray(oc[key]) - pc[key])) <= 1e-10, (name, tag)
                assert np.max(np.abs(np.asarray(ocn[key]) - pcn[key])) <= 1e-10, (name, tag, "noise")
            pm2 = gp.posterior_mean(xp, hyperparameters=thetas[0], **kw)
            fx["pm_theta1"] = np.asarray(pm2["m(x)"])
            check(name + ".pm_theta1", o.posterior_mean(xp, hyperparameters=thetas[0], **kw)["m(x)"], pm2["m(x)"], 1e-9)
        return fx

    def sweep(theta):
        return [theta * (1.0 + 0.02 * t) for t in (1, 2, 3)]

    # ---- G1 / C1: N=500 d=1 RBF -------------------------------------------------------
    x, y = synth(500, 1)
    nv = np.full(500, 0.01)
    th = np.array([1.0, 0.2])
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf,
                 kernel_function_grad=orc.rbf_ard_grad)
    o = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    fx = common(gp, o, x, y, nv, th, sweep(th), 16, 1, ref_rbf, "G1")
    # the reference's own FD kernel-gradient route (gp_prior.py:438-447) as a loose cross-check of
    # the derived RBF gradient
    gp_fd = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf)
    fx["grad_fd"] = gp_fd.neg_log_likelihood_gradient(th)
    assert rel(fx["grad"], fx["grad_fd"]) < 1e-3, (fx["grad"], fx["grad_fd"])
    np.savez_compressed(os.path.join(OUT, "G1_rbf_n500_d1.npz"), kernel="rbf_ard", **fx)

    # ---- appendix-A anchor (seed 0) --------------------------------------------------
    rng = np.random.default_rng(0)
    xa = rng.random((500, 1))
    ya = np.sin(5 * xa[:, 0]) + 0.05 * rng.standard_normal(500)
    gpa = fvgp.GP(xa, ya, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf,
                  kernel_function_grad=orc.rbf_ard_grad)
    np.savez_compressed(os.path.join(OUT, "G0_appendixA.npz"), kernel="rbf_ard", x=xa, y=ya, noise_variances=nv,
                        theta=th, loglik=gpa.log_likelihood(th), grad=gpa.neg_log_likelihood_gradient(th))

    # ---- G2: N=512 d=3 RBF + posterior ---------------------------------------------------
    x, y = synth(512, 3)
    nv = np.full(512, 0.01)
    th = np.array([1.0, 0.3, 0.3, 0.3])
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf,
                 kernel_function_grad=orc.rbf_ard_grad)
    o = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    fx = common(gp, o, x, y, nv, th, sweep(th), 16, 2, ref_rbf, "G2")
    np.savez_compressed(os.path.join(OUT, "G2_rbf_n512_d3.npz"), kernel="rbf_ard", **fx)

    # ---- G3: N=512 d=3 Matern-5/2 (gp_bo._surrogate_kernel) ---------------------------------
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=gp_bo._surrogate_kernel,
                 kernel_function_grad=gp_bo._surrogate_kernel_grad)
    o = orc.OracleGP(x, y, th, nv, kernel="matern52_ard")
    fx = common(gp, o, x, y, nv, th, sweep(th), 16, 3, gp_bo._surrogate_kernel, "G3")
    np.savez_compressed(os.path.join(OUT, "G3_matern52_n512_d3.npz"), kernel="matern52_ard", **fx)

    # ---- G4: default kernel (Matern-3/2 ARD) N=256 d=2, default noise ------------------------
    x, y = synth(256, 2)
    th = np.array([1.2, 0.4, 0.25])
    gp = fvgp.GP(x, y, init_hyperparameters=th)
    o = orc.OracleGP(x, y, th, None, kernel="matern32_ard")
    fx = common(gp, o, x, y, np.zeros(0), th, sweep(th), 16, 4, None, "G4")
    np.savez_compressed(os.path.join(OUT, "G4_default_n256_d2.npz"), kernel="matern32_ard", **fx)

    # ---- G5: fvGP, 4 tasks x 64 points d=2, default kernel -------------------------------------
    rng = np.random.default_rng(20240505)
    xm = rng.random((64, 2))
    s = np.sum(xm, axis=1)
    ym = np.stack([np.sin(3 * s), np.cos(3 * s), np.linalg.norm(xm, axis=1), np.sin(3 * s) * np.cos(3 * s)], axis=1)
    ym = ym + 0.05 * rng.standard_normal(ym.shape)
    ym_nan = ym.copy()
    ym_nan[5, 1] = np.nan
    ym_nan[17, 3] = np.nan
    nvm = np.full(ym.shape, 0.01)
    th5 = np.array([1.0, 0.3, 0.3, 1.0])
    x_out = np.arange(4.0)
    for tag, yy in (("G5_fvgp_4x64", ym), ("G5n_fvgp_4x64_nan", ym_nan)):
        gp = fvgp.fvGP(xm, yy, init_hyperparameters=th5, noise_variances=nvm)
        xt, yt, vt = orc.transform_index_set(xm, yy, nvm)
        assert np.array_equal(xt, gp.x_data) and np.array_equal(yt, gp.y_data[:, 0]) and np.array_equal(vt, gp.likelihood.V)
        o = orc.OracleGP(xt, yt, th5, vt, kernel="matern32_ard", x_out=x_out)
        fx = common(gp, o, xt, yt, vt, th5, sweep(th5), 8, 5, None, tag, x_out=x_out)
        fx.update(fvgp_x=xm, fvgp_y=yy, fvgp_noise=nvm, x_out=x_out)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), kernel="matern32_ard", **fx)

    # ---- G6: two-column y (checks /ncols, tiled variance) ---------------------------------------
    x, y2 = synth(300, 3, ncol=2)
    nv = np.full(300, 0.02)
    th = np.array([0.9, 0.35, 0.3, 0.4])
    gp = fvgp.GP(x, y2, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf,
                 kernel_function_grad=orc.rbf_ard_grad)
    o = orc.OracleGP(x, y2, th, nv, kernel="rbf_ard")
    fx = common(gp, o, x, y2, nv, th, sweep(th), 12, 6, ref_rbf, "G6", grad=True)
    fx["grad_c1"] = gp.neg_log_likelihood_gradient(th, component=1)
    check("G6.grad_c1", o.neg_log_likelihood_gradient(th, component=1), fx["grad_c1"], 1e-9)
    np.savez_compressed(os.path.join(OUT, "G6_rbf_2col_n300_d3.npz"), kernel="rbf_ard", **fx)

    # ---- G7: non positive definite -> info ----------------------------------------------------
    rng = np.random.default_rng(7)
    B = rng.standard_normal((96, 96))
    M = B @ B.T + 96 * np.eye(96)
    bad = 70
    M[bad, bad] = -1.0
    try:
        gp_lin_alg.calculate_Chol_factor(M)
        raise SystemExit("reference accepted an indefinite matrix?")
    except gp_lin_alg.NonPositiveDefiniteError as e:
        msg = str(e)
        assert isinstance(e, np.linalg.LinAlgError)
    try:
        orc.calculate_Chol_factor(M)
        raise SystemExit("oracle accepted an indefinite matrix?")
    except orc.NonPositiveDefiniteError:
        pass
    Mok = B @ B.T + 96 * np.eye(96)
    Lok = np.tril(gp_lin_alg.calculate_Chol_factor(Mok))
    rhs = rng.standard_normal((96, 3))
    np.savez_compressed(os.path.join(OUT, "G7_nonpd.npz"), M=M, info=bad + 1, message=msg,
                        Mok=Mok, Lok=Lok, rhs=rhs, sol=gp_lin_alg.calculate_Chol_solve(Lok, rhs),
                        logdet=gp_lin_alg.calculate_Chol_logdet(Lok))

    # ---- G8: isotropic notebook kernels + raw distance / radial functions -------------------------
    x, y = synth(200, 2)
    nv = np.full(200, 0.01)
    th = np.array([1.1, 0.35])
    fx8 = {"x": x, "y": y, "noise_variances": nv, "theta": th}
    for nm, rf in (("rbf_iso", ref_rbf_iso), ("matern32_iso", ref_m32_iso), ("matern52_iso", ref_m52_iso)):
        gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=rf)
        o = orc.OracleGP(x, y, th, nv, kernel=nm)
        check("G8." + nm + ".K", o.K, gp.K, 1e-14)
        fx8[nm + "_loglik"] = gp.log_likelihood(th)
        fx8[nm + "_K_corner"] = gp.K[:8, :8].copy()
        fx8[nm + "_K_fro"] = np.linalg.norm(gp.K)
        check("G8." + nm + ".loglik", o.log_likelihood(th), fx8[nm + "_loglik"], 1e-12)
    x1, x2 = x[:24], x[100:140]
    ls = np.array([0.3, 0.7])
    fx8.update(x1=x1, x2=x2, lengths=ls,
               dist_iso=rk.get_distance_matrix(x1, x2), dist_aniso=rk.get_anisotropic_distance_matrix(x1, x2, ls))
    dd = fx8["dist_aniso"]
    fx8.update(sqexp=rk.squared_exponential_kernel(dd, 0.8), mat1=rk.matern_kernel_diff1(dd, 0.8),
               mat2=rk.matern_kernel_diff2(dd, 0.8))
    check("G8.dist_iso", orc.get_distance_matrix(x1, x2), fx8["dist_iso"], 0.0)
    check("G8.dist_aniso", orc.get_anisotropic_distance_matrix(x1, x2, ls), fx8["dist_aniso"], 0.0)
    check("G8.sqexp", orc.squared_exponential_kernel(dd, 0.8), fx8["sqexp"], 0.0)
    check("G8.mat1", orc.matern_kernel_diff1(dd, 0.8), fx8["mat1"], 0.0)
    check("G8.mat2", orc.matern_kernel_diff2(dd, 0.8), fx8["mat2"], 0.0)
    th3 = np.array([1.3, 0.3, 0.7])
    fx8.update(theta3=th3,
               grad_m32=fvgp.gp_prior.GPprior._default_kernel_analytical_gradient(x1, x2, th3),
               grad_m52=gp_bo._surrogate_kernel_grad(x1, x2, th3))
    check("G8.grad_m32", orc.matern32_ard_grad(x1, x2, th3), fx8["grad_m32"], 0.0)
    check("G8.grad_m52", orc.matern52_ard_grad(x1, x2, th3), fx8["grad_m52"], 1e-15)
    fd = orc.fd_kernel_grad(orc.rbf_ard, x1, x2, th3, eps=1e-6)
    assert rel(orc.rbf_ard_grad(x1, x2, th3), fd) < 1e-6
    np.savez_compressed(os.path.join(OUT, "G8_iso_and_units.npz"), **fx8)

    # ---- G9: the finite-difference derivatives built on the path --------------------------------------
    # posterior_mean_grad / posterior_covariance_grad (gp_posterior.py:184-226,290-331),
    # neg_log_likelihood_hessian / test_log_likelihood_gradient (gp_marginal_likelihood.py:312-364)
    x, y = synth(256, 2)
    nv = np.full(256, 0.01)
    th = np.array([1.1, 0.35, 0.45])
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf,
                 kernel_function_grad=orc.rbf_ard_grad)
    o = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    xp = np.random.default_rng(99).random((10, 2))
    th2 = th * 1.03
    fx9 = {"x": x, "y": y, "noise_variances": nv, "theta": th, "theta2": th2, "x_pred": xp}
    for tag, a, b in (("dm_all", gp.posterior_mean_grad(xp)["dm/dx"], o.posterior_mean_grad(xp)["dm/dx"]),
                      ("dm_dir1", gp.posterior_mean_grad(xp, direction=1)["dm/dx"], o.posterior_mean_grad(xp, direction=1)["dm/dx"]),
                      ("dm_theta2", gp.posterior_mean_grad(xp, hyperparameters=th2)["dm/dx"],
                       o.posterior_mean_grad(xp, hyperparameters=th2)["dm/dx"]),
                      ("dv_all", gp.posterior_covariance_grad(xp)["dv/dx"], o.posterior_covariance_grad(xp)["dv/dx"]),
                      ("dv_dir0", gp.posterior_covariance_grad(xp, direction=0)["dv/dx"],
                       o.posterior_covariance_grad(xp, direction=0)["dv/dx"]),
                      ("dS_dir0", gp.posterior_covariance_grad(xp, direction=0)["dS/dx"],
                       o.posterior_covariance_grad(xp, direction=0)["dS/dx"])):
        fx9[tag] = np.asarray(a)
        check("G9." + tag, b, a, 1e-9)
    fx9["hessian"] = gp.marginal_likelihood.neg_log_likelihood_hessian(hyperparameters=th)
    check("G9.hessian", o.neg_log_likelihood_hessian(th), fx9["hessian"], 1e-9)
    fd_g, an_g = gp.test_log_likelihood_gradient(th)
    fx9["fd_grad"], fx9["an_grad"] = fd_g, an_g
    ofd, oan = o.test_log_likelihood_gradient(th)
    check("G9.fd_grad", ofd, fd_g, 1e-9)
    check("G9.an_grad", oan, an_g, 1e-9)
    np.savez_compressed(os.path.join(OUT, "G9_derivatives_rbf_n256_d2.npz"), kernel="rbf_ard", **fx9)
    # multi-task shapes of the same derivatives (x_out reshapes, order='F')
    gp = fvgp.fvGP(xm, ym, init_hyperparameters=th5, noise_variances=nvm)
    xt, yt, vt = orc.transform_index_set(xm, ym, nvm)
    o = orc.OracleGP(xt, yt, th5, vt, kernel="matern32_ard", x_out=x_out)
    xp5 = np.random.default_rng(98).random((6, 2))
    fx9m = {"fvgp_x": xm, "fvgp_y": ym, "fvgp_noise": nvm, "theta": th5, "x_pred": xp5, "x_out": x_out}
    for tag, a, b in (("dm_all", gp.posterior_mean_grad(xp5, x_out=x_out)["dm/dx"], o.posterior_mean_grad(xp5, x_out=x_out)["dm/dx"]),
                      ("dm_dir0", gp.posterior_mean_grad(xp5, x_out=x_out, direction=0)["dm/dx"],
                       o.posterior_mean_grad(xp5, x_out=x_out, direction=0)["dm/dx"]),
                      ("dv_all", gp.posterior_covariance_grad(xp5, x_out=x_out)["dv/dx"],
                       o.posterior_covariance_grad(xp5, x_out=x_out)["dv/dx"]),
                      ("dv_dir1", gp.posterior_covariance_grad(xp5, x_out=x_out, direction=1)["dv/dx"],
                       o.posterior_covariance_grad(xp5, x_out=x_out, direction=1)["dv/dx"]),
                      ("dS_dir1", gp.posterior_covariance_grad(xp5, x_out=x_out, direction=1)["dS/dx"],
                       o.posterior_covariance_grad(xp5, x_out=x_out, direction=1)["dS/dx"])):
        fx9m[tag] = np.asarray(a)
        check("G9m." + tag, b, a, 1e-9)
    np.savez_compressed(os.path.join(OUT, "G9m_derivatives_fvgp_4x64.npz"), kernel="matern32_ard", **fx9m)

    # ---- G10: validation scores and P x P information measures (callers of the posterior) ----------------
    x, y = synth(400, 2)
    nv = np.full(400, 0.01)
    th = np.array([1.1, 0.35, 0.45])
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=ref_rbf)
    o = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    rng = np.random.default_rng(77)
    xt = rng.random((40, 2))
    yt = np.sin(3.0 * np.sum(xt, axis=1)) + 0.1 * rng.standard_normal(40)
    ref_scores = {"rmse": gp.rmse(xt, yt), "nrmse": gp.nrmse(xt, yt), "mae": gp.mae(xt, yt), "mape": gp.mape(xt, yt),
                  "r2": gp.r2(xt, yt), "nlpd": gp.nlpd(xt, yt), "msll": gp.msll(xt, yt),
                  "crps_mean": gp.crps(xt, yt)[0], "crps_std": gp.crps(xt, yt)[1], "picp": gp.picp(xt, yt),
                  "mpiw": gp.mpiw(xt), "interval_score": gp.interval_score(xt, yt)}
    osc = orc.validation_scores(o, xt, yt)
    for k2, v2 in ref_scores.items():
        check("G10." + k2, osc[k2], v2, 1e-10)
    cc = gp.coverage_curve(xt, yt)
    xq = rng.random((6, 2))
    cm = rng.standard_normal(6) * 0.1 + np.sin(3.0 * np.sum(xq, axis=1))
    Bq = rng.standard_normal((6, 6))
    cq = Bq @ Bq.T * 0.01 + 0.05 * np.eye(6)
    ref_info = {"kl_div": gp.gp_kl_div(xq, cm, cq)["kl-div"], "rie": gp.gp_relative_information_entropy(xq)["RIE"],
                "rie_set": gp.gp_relative_information_entropy_set(xq)["RIE"]}
    pp = gp.posterior_probability(xq, cm, cq)
    ref_info.update(pp_mu=pp["mu"], pp_cov=pp["covariance"], pp_prob=pp["probability"])
    oi = orc.information_measures(o, xq, cm, cq)
    for k2, v2 in ref_info.items():
        check("G10." + k2, oi[k2], v2, 1e-8)
    np.savez_compressed(os.path.join(OUT, "G10_scores_rbf_n400_d2.npz"), kernel="rbf_ard", x=x, y=y, noise_variances=nv, theta=th,
                        x_test=xt, y_test=yt, x_q=xq, comp_mean=cm, comp_cov=cq,
                        coverage_target=np.array(cc["target_coverage"]), coverage_measured=np.array(cc["measured_coverage"]),
                        grid2d=fvgp.GP.make_2d_x_pred([0, 1], [2, 3], 4, 3), grid1d=fvgp.GP.make_1d_x_pred([0, 2], 5),
                        **{"score_" + k2: v2 for k2, v2 in ref_scores.items()}, **{"info_" + k2: v2 for k2, v2 in ref_info.items()})

    # ---- G11: seeded training traces: the MCMC chain (numpy's legacy global stream, gp_mcmc.py:214,337-341) and the
    #      Adam history (gp_training.py:576-667) of the reference, Matern-5/2 with its analytic gradient ---------------
    x, y = synth(200, 2, seed=11)
    nv = np.full(200, 0.01)
    th = np.array([1.0, 0.4, 0.5])
    bounds = np.array([[0.1, 5.0], [0.05, 2.0], [0.05, 2.0]])
    gp = fvgp.GP(x, y, init_hyperparameters=th, noise_variances=nv, kernel_function=gp_bo._surrogate_kernel,
                 kernel_function_grad=gp_bo._surrogate_kernel_grad)
    np.random.seed(20240917)
    hps_mcmc = gp.train(hyperparameter_bounds=bounds, init_hyperparameters=th, method="mcmc", max_iter=150)
    info = gp.trainer.mcmc_info
    gp.set_hyperparameters(th)
    th_adam, hist = gp.trainer.adam_optimize(gp.marginal_likelihood.neg_log_likelihood, gp.marginal_likelihood.neg_log_likelihood_gradient, th, max_iter=30)
    o = orc.OracleGP(x, y, th, nv, kernel="matern52_ard")
    check("G11.loglik_on_chain", [o.log_likelihood(t) for t in np.asarray(info["x"])[1::25]], np.asarray(info["f(x)"])[0::25][:len(np.asarray(info["x"])[1::25])], 1e-11)
    np.savez_compressed(os.path.join(OUT, "G11_training_traces_m52_n200_d2.npz"), kernel="matern52_ard", x=x, y=y, noise_variances=nv,
                        theta=th, bounds=bounds, seed=20240917, mcmc_max_iter=150, mcmc_x=np.asarray(info["x"]),
                        mcmc_f=np.asarray(info["f(x)"]), mcmc_median=np.asarray(info["median(x)"]), mcmc_result=hps_mcmc,
                        adam_max_iter=30, adam_theta=np.asarray(hist["theta"]), adam_nlml=np.asarray(hist["nlml"]),
                        adam_grad_norm=np.asarray(hist["grad_norm"]), adam_result=th_adam)

    print(f"{'check':32s} {'rel.diff':>10s} {'tol':>8s}")
    for tag, r, tol in report:
        print(f"{tag:32s} {r:10.2e} {tol:8.0e}")
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
