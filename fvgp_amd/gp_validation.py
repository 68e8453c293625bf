"""Validation scores and P x P information measures on top of the posterior (callers of the path).

Everything here is O(P) .. O(P^3) host arithmetic on the dictionaries `posterior_mean` /
`posterior_covariance` return (fvgp/gp.py:1754-2071; gp_posterior.py:391-552), so the device work is exactly
one mean and one covariance evaluation per score.  The (N+P)-sized joint-prior measures of the reference
(gp_entropy, gp_mutual_information, gp_total_correlation) are not provided.
"""
import warnings

import numpy as np
from scipy.stats import norm


def _same_shape(name, *arrays):
    shapes = [np.shape(a) for a in arrays]
    assert all(s == shapes[0] for s in shapes), f"{name}: shape mismatch {shapes}"


def _logdet(S):
    sign, ld = np.linalg.slogdet(S)          # gp_lin_alg.calculate_logdet (:1484-1540), dense branch
    return ld


class ValidationMixin:
    """Mixed into fvgp_amd.GP: needs posterior_mean, posterior_covariance, y_data, x_out, _mean, _hps."""

    # -- helpers -----------------------------------------------------------------------------------
    def _mean_at(self, x_test):
        return self.posterior_mean(x_test)["m(x)"]

    def _var_at(self, x_test, add_noise=False):
        return self.posterior_covariance(x_test, add_noise=add_noise)["v(x)"]

    @staticmethod
    def _z(interval):
        return norm.ppf(1.0 - (1.0 - interval) / 2.0)

    # -- point scores (gp.py:1784-1874,1994-2038) ---------------------------------------------------------
    def rmse(self, x_test, y_test):
        m = self._mean_at(x_test)
        _same_shape("rmse", y_test, m)
        return np.sqrt(np.sum((y_test - m) ** 2) / y_test.size)

    def nrmse(self, x_test, y_test):
        return self.rmse(x_test, y_test) / (np.max(y_test) - np.min(y_test))

    def mae(self, x_test, y_test):
        m = self._mean_at(x_test)
        _same_shape("mae", y_test, m)
        return np.mean(np.abs(y_test - m))

    def mape(self, x_test, y_test):
        m = self._mean_at(x_test)
        _same_shape("mape", y_test, m)
        return np.mean(np.abs((y_test - m) / y_test))

    def r2(self, x_test, y_test):
        m = self._mean_at(x_test)
        _same_shape("r2", y_test, m)
        return 1.0 - np.sum((y_test - m) ** 2) / np.sum((y_test - np.mean(y_test)) ** 2)

    # -- density scores (gp.py:1754-1782,1827-1852,2040-2071) ----------------------------------------------
    @staticmethod
    def _gaussian_nlpd(y, mean, var):
        return np.mean(0.5 * np.log(2.0 * np.pi * var) + 0.5 * (y - mean) ** 2 / var)

    def nlpd(self, x_test, y_test):
        m, v = self._mean_at(x_test), self._var_at(x_test)
        _same_shape("nlpd", y_test, m, v)
        return self._gaussian_nlpd(y_test, m, v)

    def msll(self, x_test, y_test):
        """NLPD of the GP minus the NLPD of the trivial N(mean(y_data), var(y_data)) predictor."""
        m, v = self._mean_at(x_test), self._var_at(x_test)
        _same_shape("msll", y_test, m, v)
        return self._gaussian_nlpd(y_test, m, v) - self._gaussian_nlpd(y_test, np.mean(self.y_data), np.var(self.y_data))

    @staticmethod
    def _crps_s(x, mu, sigma):
        t = (x - mu) / sigma
        res = abs(sigma * (1.0 / np.sqrt(np.pi) - 2.0 * norm.pdf(t) - t * (2.0 * norm.cdf(t) - 1.0)))
        return np.mean(res), np.sqrt(np.var(res))

    def crps(self, x_test, y_test):
        """(mean, standard deviation) of the continuous ranked probability score."""
        m, s = self._mean_at(x_test), np.sqrt(self._var_at(x_test))
        _same_shape("crps", y_test, m, s)
        return self._crps_s(y_test, m, s)

    # -- interval scores, with the noise added to the variance (gp.py:1876-1992) ---------------------------------
    def picp(self, x_test, y_true, interval=0.95):
        m, s = self._mean_at(x_test), np.sqrt(self._var_at(x_test, add_noise=True))
        z = self._z(interval)
        return np.mean((y_true >= m - z * s) & (y_true <= m + z * s))

    def coverage_curve(self, x_test, y_test, intervals=None):
        if intervals is None:
            intervals = np.linspace(0.05, 0.95, 19)
        return {"target_coverage": list(intervals),
                "measured_coverage": [self.picp(x_test, y_test, interval=q) for q in intervals]}

    def mpiw(self, x_test, interval=0.95):
        s = np.sqrt(np.clip(self._var_at(x_test, add_noise=True), 0.0, None))
        return np.mean(2.0 * self._z(interval) * s)

    def interval_score(self, x_test, y_test, interval=0.95):
        m, s = self._mean_at(x_test), np.sqrt(self._var_at(x_test, add_noise=True))
        _same_shape("interval_score", y_test, m, s)
        a = 1.0 - interval
        z = norm.ppf(1.0 - a / 2.0)
        lo, hi = m - z * s, m + z * s
        return np.mean((hi - lo) + (2.0 / a) * np.maximum(lo - y_test, 0.0) + (2.0 / a) * np.maximum(y_test - hi, 0.0))

    # -- small helpers of the reference's public surface (gp.py:2130-2185) ------------------------------------
    @staticmethod
    def gaussian_1d(x, mu, sigma):
        return np.exp(-((x - mu) ** 2) / (2.0 * sigma ** 2)) / (np.sqrt(2.0 * np.pi) * sigma)

    @staticmethod
    def make_1d_x_pred(b, res=100):
        return np.linspace(b[0], b[1], res).reshape(res, -1)

    @staticmethod
    def make_2d_x_pred(bx, by, resx=100, resy=100):
        gx, gy = np.meshgrid(np.linspace(bx[0], bx[1], resx), np.linspace(by[0], by[1], resy), indexing="ij")
        return np.stack([gx.ravel(), gy.ravel()], axis=1)        # same order as itertools.product(x, y)

    # -- information measures between P-dimensional normals (gp_posterior.py:391-552) ----------------------------
    @staticmethod
    def entropy(S):
        dim = len(S[0])
        return 0.5 * dim * (1.0 + np.log(2.0 * np.pi)) + 0.5 * _logdet(S)

    @staticmethod
    def kl_div(mu1, mu2, S1, S2):
        """KL(N(mu1,S1) || N(mu2,S2)); the reference returns its absolute value and warns below -1e-4."""
        dmu = np.subtract(mu2, mu1)
        kld = 0.5 * (np.trace(np.linalg.solve(S2, S1)) + dmu @ np.linalg.solve(S2, dmu) - float(len(dmu))
                     + (_logdet(S2) - _logdet(S1)))
        if kld < -1e-4:
            warnings.warn("Negative KL divergence encountered. That happens when one of the covariance matrices is "
                          "close to positive semi definite and therefore the logdet() calculation becomes unstable. "
                          "Returning abs(KLD)")
        return abs(kld)

    def gp_kl_div(self, x_pred, comp_mean, comp_cov, x_out=None):
        if x_out is None:
            x_out = self.x_out
        gp_mean = self.posterior_mean(x_pred, x_out=x_out)["m(x)_flat"]
        gp_cov = self.posterior_covariance(x_pred, x_out=x_out)["S_flat"]
        gp_cov = gp_cov + np.identity(len(gp_cov)) * 1e-9
        comp_cov = comp_cov + np.identity(len(comp_cov)) * 1e-9
        return {"x": x_pred, "gp posterior mean": gp_mean, "gp posterior covariance": gp_cov,
                "given mean": comp_mean, "given covariance": comp_cov,
                "kl-div": self.kl_div(gp_mean, comp_mean, gp_cov, comp_cov)}

    def gp_relative_information_entropy(self, x_pred, x_out=None, add_noise=False):
        """KL(prior || posterior) on the prediction points."""
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        x_aux = self.cartesian_product(x_pred, x_out) if isinstance(x_out, np.ndarray) else x_pred
        kk = self._kk_host(x_aux, self._hps) + np.identity(len(x_aux)) * 1e-9
        post_cov = self.posterior_covariance(x_pred, x_out=x_out, add_noise=add_noise)["S_flat"]
        post_cov = post_cov + np.identity(len(post_cov)) * 1e-9
        post_mean = self.posterior_mean(x_pred, x_out=x_out)["m(x)_flat"]
        return {"x": x_pred.copy(), "RIE": self.kl_div(self._mean(x_aux, self._hps), post_mean, kk, post_cov)}

    def gp_relative_information_entropy_set(self, x_pred, x_out=None, add_noise=False):
        rie = np.array([self.gp_relative_information_entropy(x_pred[i].reshape(1, -1), x_out=x_out, add_noise=add_noise)["RIE"]
                        for i in range(len(x_pred))])
        return {"x": x_pred.copy(), "RIE": rie}

    def posterior_probability(self, x_pred, comp_mean, comp_cov, x_out=None):
        """Probability of the product of the (noisy) GP posterior and a given normal (gp_posterior.py:527-552)."""
        if x_out is None:
            x_out = self.x_out
        self._perform_input_checks(x_pred, x_out)
        g_mean = self.posterior_mean(x_pred, x_out=x_out)["m(x)_flat"]
        g_cov = self.posterior_covariance(x_pred, x_out=x_out, add_noise=True)["S_flat"]
        gi, ci = np.linalg.inv(g_cov), np.linalg.inv(comp_cov)
        cov = np.linalg.inv(gi + ci)
        w = gi @ g_mean + ci @ comp_mean
        mu = cov @ w
        c = 0.5 * (w @ cov @ w - (g_mean @ gi @ g_mean + comp_mean @ ci @ comp_mean))
        dim = len(mu)
        ln_p = (c + 0.5 * _logdet(cov)) - (np.log((2.0 * np.pi) ** (dim / 2.0)) + 0.5 * (_logdet(g_cov) + _logdet(comp_cov)))
        return {"mu": mu, "covariance": cov, "probability": np.exp(ln_p)}
