"""Training drivers -- the callers of the hot path.  Host-side optimisers around a
device-resident objective: every objective call is one fused evaluation in HBM, theta in and a
scalar (or (H,) gradient) out.  Mirrors the Dask-free methods of GPtraining.train
(fvgp/gp_training.py:28-196): 'mcmc' (the default, gp_mcmc.py:96-224), 'adam' (:576-667), 'global', 'local'.
"""
import warnings

import numpy as np


def _in_bounds(theta, bounds):
    return bool(np.all((theta >= bounds[:, 0]) & (theta <= bounds[:, 1])))


class ProposalDistribution:
    """A user-defined proposal distribution over some of the hyperparameters (gp_mcmc.py:234-364, same constructor):
    `proposal_dist(x_part, x_all, obj)` returns the proposal for the entries `indices` ("normal": N(x_part,
    prop_args["prop_Sigma"])), `adapt_callable(iteration, chain)` may update `obj.prop_args` (chain.trace["x"] is the list
    of positions so far; "normal": the adaptive covariance of :337-356), auto_accept skips the Metropolis test."""

    def __init__(self, indices, proposal_dist="normal", init_prop_Sigma=None, adapt_callable=None, r_opt=.234, c_0=10, c_1=.8,
                 K=10, auto_accept=False, adapt_cov=True, prop_args=None, ID=None):
        self.indices, self.r_opt, self.c_0, self.c_1, self.K = indices, r_opt, c_0, c_1, K
        self.auto_accept, self.adapt_cov, self.ID, self.jump_trace = auto_accept, adapt_cov, ID, []
        dim = len(indices)
        if proposal_dist == "normal":
            self.proposal_dist = self.normal_proposal_dist
        elif callable(proposal_dist):
            self.proposal_dist = proposal_dist
        else:
            raise Exception("No proposal distribution specified!")
        if proposal_dist == "normal" and init_prop_Sigma is None:
            init_prop_Sigma = np.identity(dim)
            warnings.warn("You are using the normal proposal distribution for normal distributions\n "
                          "but did not provide `init_prop_sigma`. This can lead to slow convergence")
        if callable(adapt_callable):
            self.adapt = adapt_callable
        elif adapt_callable == "normal" or proposal_dist == "normal":
            self.adapt = self._adapt
        else:
            if isinstance(adapt_callable, str):
                raise Exception("Invalid string provided for adapt callable.")
            self.adapt = lambda end, chain: None
        if prop_args is None:
            self.prop_args = {"prop_Sigma": init_prop_Sigma, "sigma_m": 2.4 ** 2 / dim}
        else:
            self.prop_args = prop_args
            if adapt_callable == "normal":
                self.prop_args["prop_Sigma"] = init_prop_Sigma
                self.prop_args["sigma_m"] = 2.4 ** 2 / dim

    def normal_proposal_dist(self, x, hps, obj):
        return obj.rng.multivariate_normal(mean=x, cov=obj.prop_args["prop_Sigma"], size=1).reshape(len(x))

    def _adapt(self, end, chain):
        K = self.K
        if (end % K) == 0:
            start = end - K + 1
            gamma2 = 1. / ((end / K) + 3) ** self.c_1
            r_hat = np.mean(self.jump_trace[start:end])
            self.prop_args["sigma_m"] = np.exp(np.log(self.prop_args["sigma_m"]) + self.c_0 * gamma2 * (r_hat - self.r_opt))
            if self.adapt_cov:
                seg = np.asarray(chain.trace["x"]).T[self.indices, start:end]
                self.prop_args["prop_Sigma"] = self.prop_args["prop_Sigma"] + gamma2 * (np.cov(seg) - self.prop_args["prop_Sigma"])

    rng = np.random


class _Chain:
    """what adapt callables and run_in_every_iteration see of the sampler (gpMCMC's `trace` and `args`)"""

    def __init__(self, args):
        self.trace, self.args = {"f(x)": [], "x": [], "time stamp": []}, args


def run_mcmc_proposals(log_likelihood, bounds, x0, proposal_distributions, n_updates=10000, info=False, rng=None, prior=None,
                       args=None, break_default=True):
    """gpMCMC.run_mcmc / _jump (gp_mcmc.py:96-224) with user ProposalDistribution objects: per iteration every proposal
    object moves its own entries in turn (one likelihood evaluation each), then adapts."""
    rng = np.random if rng is None else rng
    if prior is None:
        def prior(theta, box, _args):
            return 0.0 if _in_bounds(theta, box) else -np.inf
    chain = _Chain(args)
    x = np.array(x0, dtype=np.float64)
    chain.trace["x"].append(x.copy())
    f = log_likelihood(x)
    p = prior(x, bounds, args)
    accepted = []
    for obj in proposal_distributions:
        obj.rng = rng
    for i in range(1, max(int(n_updates), 2)):
        for obj in proposal_distributions:
            x_star = x.copy()
            x_star[obj.indices] = obj.proposal_dist(x[obj.indices].copy(), x, obj)
            p_star, jumped = prior(x_star, bounds, args), 0.0
            if p_star != -np.inf:
                f_star = log_likelihood(x_star)
                if np.isnan(f_star):
                    raise Exception("Likelihood evaluation = NaN in gpMCMC")
                expo = p_star + f_star - p - f
                ratio = np.exp(expo) if expo < 50 else 1.1
                if np.isnan(ratio):
                    ratio = 0.0
                if ratio > rng.uniform(0, 1, 1) or obj.auto_accept:
                    x, f, p, jumped = x_star, f_star, p_star, 1.0
            obj.jump_trace.append(jumped)
            accepted.append(jumped)
            obj.adapt(i, chain)
        chain.trace["x"].append(x.copy())
        chain.trace["f(x)"].append(f)
        if info and i % 10 == 0:
            print("Finished ", i, " out of ", n_updates, " iterations. f(x)= ", f)
        if break_default and len(chain.trace["f(x)"]) >= 1000:
            fl = np.asarray(chain.trace["f(x)"])
            if abs(fl[-100:].mean() - fl[-200:-100].mean()) < 1e-3:
                break
    xs, fs = np.asarray(chain.trace["x"]), chain.trace["f(x)"]
    dist_index = int(len(xs) - (len(xs) / 100))
    arg_max = int(np.argmax(fs))
    return {"f(x)": fs, "max f(x)": fs[arg_max], "MAP": fs[arg_max], "max x": xs[arg_max], "x": xs,
            "mean(x)": np.mean(xs[dist_index:], axis=0), "median(x)": np.median(xs[dist_index:], axis=0),
            "var(x)": np.var(xs[dist_index:], axis=0), "acceptance": float(np.mean(accepted)) if accepted else 0.0}


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
          objective_function_gradient=None, objective_function_hessian=None, mcmc_prior=None, mcmc_args=None,
          mcmc_prop_distrs="normal"):
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
        rng = None if seed is None else np.random.RandomState(seed)
        if mcmc_prop_distrs in ("normal", None):
            res = run_mcmc(objective_function, bounds, init_hyperparameters, n_updates=max_iter, info=info, rng=rng,
                           prior=mcmc_prior, args=mcmc_args)
        else:                                                                   # the user's ProposalDistribution objects
            res = run_mcmc_proposals(objective_function, bounds, init_hyperparameters, list(mcmc_prop_distrs), n_updates=max_iter,
                                     info=info, rng=rng, prior=mcmc_prior, args=mcmc_args)
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
