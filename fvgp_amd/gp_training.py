"""Training drivers -- the callers of the hot path.  Host-side optimisers around a
device-resident objective: every objective call is one fused evaluation in HBM, theta in and a
scalar (or (H,) gradient) out.  Mirrors the Dask-free methods of GPtraining.train
(fvgp/gp_training.py:28-196): 'mcmc' (the default, gp_mcmc.py:96-224), 'adam' (:576-667), 'global', 'local'.
"""
import warnings

import numpy as np


def _in_bounds(theta, bounds):
    return bool(np.all((theta >= bounds[:, 0]) & (theta <= bounds[:, 1])))


def run_mcmc(log_likelihood, bounds, x0, n_updates=10000, info=False, rng=None, break_default=True, prior=None,
             args=None):
    """Adaptive Metropolis-Hastings with one normal proposal over all hyperparameters.

    Follows gpMCMC.run_mcmc/_jump (gp_mcmc.py:96-224) and ProposalDistribution._adapt (:337-356):
    uniform prior on the bounds box, initial proposal covariance diag((0.2*range/sqrt(12))^2)
    (:84-87), covariance adapted every K=10 steps with gamma2 = 1/(i/K+3)^0.8, and the
    'default' break condition (|mean of last 100 f - mean of previous 100| < 1e-3 after 1000
    iterations, :181-190).  One likelihood evaluation per proposal.  `prior(theta, bounds, args)` is a log prior
    (-inf = reject without an evaluation, gp_mcmc.py:203-209); the default is the uniform box.  Returns the
    reference's info dict; GP.train takes "median(x)" = median of the last 1 % of the trace.
    """
    # The reference draws from numpy's legacy global stream (np.random.multivariate_normal / np.random.uniform,
    # gp_mcmc.py:214,337-341): the default here is that same module, so `np.random.seed(s)` before train() walks the
    # reference's chain; a RandomState or a Generator may be passed instead.  The two calls below have the
    # reference's exact argument shapes, which is what keeps a seeded chain bit-identical to its.
    rng = np.random if rng is None else rng
    n_updates = max(int(n_updates), 2)
    dim = len(bounds)
    std = (bounds[:, 1] - bounds[:, 0]) * 0.2 / np.sqrt(12)
    prop_Sigma = np.diag(std ** 2)
    K, c_1 = 10, 0.8
    if prior is None:
        def prior(theta, box, _args):
            return 0.0 if _in_bounds(theta, box) else -np.inf
    x = np.array(x0, dtype=np.float64)
    f = log_likelihood(x)
    p = prior(x, bounds, args)
    trace_x, trace_f, jumps = [x.copy()], [], []
    for i in range(1, n_updates):
        x_star = rng.multivariate_normal(mean=x, cov=prop_Sigma, size=1).reshape(len(x))
        jumped = 0.0
        p_star = prior(x_star, bounds, args)
        if p_star != -np.inf:
            f_star = log_likelihood(x_star)
            if np.isnan(f_star):
                raise Exception("Likelihood evaluation = NaN in gpMCMC")
            expo = p_star + f_star - p - f
            ratio = np.exp(expo) if expo < 50 else 1.1
            if np.isnan(ratio):
                ratio = 0.0
            if ratio > rng.uniform(0, 1, 1):
                x, f, p, jumped = x_star, f_star, p_star, 1.0
        jumps.append(jumped)
        if i % K == 0:
            start = i - K + 1
            gamma2 = 1.0 / ((i / K) + 3) ** c_1
            seg = np.asarray(trace_x).T[:, start:i]
            if seg.shape[1] > 1:
                prop_Sigma = prop_Sigma + gamma2 * (np.atleast_2d(np.cov(seg)) - prop_Sigma)
        trace_x.append(x.copy())
        trace_f.append(f)
        if info and i % 10 == 0:
            print("Finished ", i, " out of ", n_updates, " iterations. f(x)= ", f)
        if break_default and len(trace_f) >= 1000:
            fl = np.asarray(trace_f)
            if abs(fl[-100:].mean() - fl[-200:-100].mean()) < 1e-3:
                break
    xs = np.asarray(trace_x)
    dist_index = int(len(xs) - (len(xs) / 100))
    arg_max = int(np.argmax(trace_f))
    return {"f(x)": trace_f, "max f(x)": trace_f[arg_max], "MAP": trace_f[arg_max], "max x": xs[arg_max], "x": xs,
            "mean(x)": np.mean(xs[dist_index:], axis=0), "median(x)": np.median(xs[dist_index:], axis=0),
            "var(x)": np.var(xs[dist_index:], axis=0), "acceptance": float(np.mean(jumps)) if jumps else 0.0}


def adam_optimize(nlml, grad_nlml, theta0, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, max_iter=1000, tol=1e-6,
                  callback=None, early_stop=None):
    """Adam on the negative log marginal likelihood (GPtraining.adam_optimize, gp_training.py:576-667): one value and
    one gradient evaluation per step, bias-corrected moments, stop when the parameter update is shorter than `tol`.
    Returns (theta, history) with history = {"theta", "nlml", "grad_norm"} as the reference keeps it."""
    theta = np.array(theta0, dtype=np.float64)
    m = np.zeros(theta.size)
    v = np.zeros(theta.size)
    history = {"theta": [], "nlml": [], "grad_norm": []}
    for t in range(1, int(max_iter) + 1):
        fval = nlml(theta)
        g = np.asarray(grad_nlml(theta), dtype=np.float64)
        m = beta1 * m + (1.0 - beta1) * g
        v = beta2 * v + (1.0 - beta2) * (g ** 2)
        m_hat = m / (1.0 - beta1 ** t)
        v_hat = v / (1.0 - beta2 ** t)
        theta_new = theta - lr * m_hat / (np.sqrt(v_hat) + eps)
        history["theta"].append(theta.copy())
        history["nlml"].append(fval)
        history["grad_norm"].append(np.linalg.norm(g))
        if callback is not None:
            callback(theta, fval, g, t)
        if np.linalg.norm(theta_new - theta) < tol or (early_stop is not None and early_stop()):
            theta = theta_new
            break
        theta = theta_new
    return theta, history


def train(gp, bounds, init_hyperparameters, method="mcmc", pop_size=20, tolerance=1e-4, max_iter=10000,
          local_optimizer="L-BFGS-B", constraints=(), info=False, seed=None, objective_function=None,
          objective_function_gradient=None, objective_function_hessian=None, mcmc_prior=None, mcmc_args=None):
    """Dispatch on `method` (GPtraining.train, fvgp/gp_training.py:28-196).  The objective is log_likelihood for
    'mcmc' and neg_log_likelihood (+ gradient) otherwise unless the caller hands in their own (gp.py:1038-1053); a
    callable `method` gets the GP and returns the hyperparameters (:194); the result must be a 1-d ndarray (:196)."""
    assert isinstance(bounds, np.ndarray) and bounds.ndim == 2 and bounds.shape[1] == 2, "wrong bounds format"
    if len(bounds) != len(init_hyperparameters):
        raise Exception("init_hyperparameters and hyperparameter_bounds have different lengths")
    if not _in_bounds(init_hyperparameters, bounds):
        raise Exception("Starting positions outside of optimization bounds.", init_hyperparameters, bounds)
    if objective_function is None and method in ("mcmc", "global", "local", "adam"):
        objective_function = gp.log_likelihood if method == "mcmc" else gp.neg_log_likelihood
    if objective_function_gradient is None and method in ("local", "adam"):
        objective_function_gradient = gp.neg_log_likelihood_gradient
    if method == "mcmc":
        res = run_mcmc(objective_function, bounds, init_hyperparameters, n_updates=max_iter, info=info,
                       rng=None if seed is None else np.random.RandomState(seed), prior=mcmc_prior, args=mcmc_args)
        gp.mcmc_info = res
        hps = res["median(x)"]
    elif method == "global":
        from scipy.optimize import differential_evolution
        res = differential_evolution(objective_function, bounds, maxiter=max_iter, popsize=pop_size, tol=tolerance,
                                     disp=info, polish=False, x0=init_hyperparameters.reshape(1, -1),
                                     constraints=constraints, workers=1, seed=seed)
        hps = np.array(res["x"])
    elif method == "local":
        from scipy.optimize import minimize
        progress = None
        if info:
            state = {"i": 0}

            def progress(intermediate_result):                               # gp_training.py:96-101
                state["i"] += 1
                print(f"fvGP local iteration {state['i']}: f(x)= {float(intermediate_result.fun)}")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize(objective_function, init_hyperparameters, method=local_optimizer,
                           jac=objective_function_gradient, hess=objective_function_hessian, bounds=bounds,
                           tol=tolerance, callback=progress, constraints=constraints, options={"maxiter": max_iter})
        hps = res["x"]
    elif method == "adam":
        progress = None
        if info:
            def progress(theta, fval, grad, iteration):                      # gp_training.py:163-177
                if iteration % 10 == 0 or iteration == 1:
                    print(f"fvGP adam iteration {iteration} out of {max_iter}: f(x)= {float(fval)}, |grad|= {float(np.linalg.norm(grad))}")
        hps, history = adam_optimize(objective_function, objective_function_gradient, init_hyperparameters,
                                     max_iter=max_iter, callback=progress)
        gp.adam_history = history
    elif method in ("hgdl", "bo"):
        raise NotImplementedError(f"train(method={method!r}) needs HGDL / the BO surrogate loop, which are outside this "
                                  "engine's scope: 'mcmc', 'adam', 'global', 'local' or a callable run here")
    elif callable(method):
        hps = method(gp)
    else:
        raise ValueError("No optimization mode specified in fvGP")
    assert isinstance(hps, np.ndarray) and np.ndim(hps) == 1, "Optimizer returned invalid hyperparameters: " + str(hps)
    return hps
