"""ShardedGP's partition / collective logic on CPU: gloo process groups of 2 and 3 ranks with the
torch-CPU stand-in ops (tests/dist_stub_ops.py), checked against the oracle -- the analogue of the
reference's in-process Dask cluster tests (tests/test_fvgp.py:3027-3062,3152-3183: distributed ==
dense).  The HIP ops take the stand-in's place on the GPU box (tests/test_gpu_dist.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth
from oracle import fvgp_oracle as orc

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from dist_stub_ops import StubOps
from fvgp_amd.dist import ShardedGP
dist.init_process_group(backend="gloo")
n, d, panel, kernel = {n}, {d}, {panel}, {kernel!r}
x, y = synth(n, d)
if {ncol} > 1:
    y = np.stack([y, np.cos(x.sum(axis=1))], axis=1)
gp = ShardedGP(x, y, np.full(n, 0.01), kernel=kernel, ops=StubOps(), panel=panel)
theta = np.array({theta})
ll, logdet, quad = gp.log_likelihood(theta)
if dist.get_rank() == 0:
    print("RESULT " + json.dumps([ll, logdet, quad]))
dist.destroy_process_group()
'''


WORKER_SOLVES = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from dist_stub_ops import StubOps
from fvgp_amd.dist import ShardedGP
dist.init_process_group(backend="gloo")
n, d, panel, kernel = {n}, {d}, {panel}, {kernel!r}
x, y = synth(n, d)
if {ncol} > 1:
    y = np.stack([y, np.cos(x.sum(axis=1))], axis=1)
gp = ShardedGP(x, y, np.full(n, 0.01), kernel=kernel, ops=StubOps(), panel=panel)
theta = np.array({theta})
ll, logdet, quad = gp.evaluate(theta, want_alpha=True)
xp = np.random.default_rng(7).random(({npred}, d))
mean, S = gp.posterior(xp)
g = gp.gradient(component={ncol} - 1, slab=256)
out = dict(ll=ll, alpha=gp.alpha[:n, :{ncol}].numpy().tolist(), mean=mean.tolist(), S=S.tolist(), g=g.tolist())
# every rank must hold the same replicated answers
chk = torch.tensor([ll, float(np.sum(mean)), float(np.sum(S)), float(np.sum(g))], dtype=torch.float64)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
out["replicated"] = bool(torch.all(lo == hi))
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


WORKER_FACADE = r'''
import os, sys, json, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
from dist_stub_ops import StubOps
import fvgp_amd
dist.init_process_group(backend="gloo")
fx = load_golden({fixture!r})
warnings.simplefilter("ignore")
gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                 kernel_function=str(fx["kernel"]), args={{"process_group": True, "shard_ops": StubOps(), "shard_panel": {panel}}})
out = dict(loglik=gp.log_likelihood(), logliks=[gp.log_likelihood(t) for t in fx["thetas"]], loglik_again=gp.log_likelihood(),
           KVinvY=gp.KVinvY.tolist(), grad=gp.neg_log_likelihood_gradient(fx["theta"]).tolist(),
           grad_cached=gp.neg_log_likelihood_gradient().tolist(),
           pm=np.asarray(gp.posterior_mean(fx["x_pred"])["m(x)"]).tolist(),
           pm_theta1=np.asarray(gp.posterior_mean(fx["x_pred"], hyperparameters=fx["thetas"][0])["m(x)"]).tolist(),
           pS=gp.posterior_covariance(fx["x_pred"])["S"].tolist(),
           pv_noise=gp.posterior_covariance(fx["x_pred"], variance_only=True, add_noise=True)["v(x)"].tolist())
# what the reference keeps as whole host arrays is gathered on request (fvgp/gp.py:625-635), and the object pickles
import pickle
from oracle import fvgp_oracle as orc
Kref = orc.KERNELS[str(fx["kernel"])](fx["x"], fx["x"], fx["theta"])
out["K_err"] = float(np.max(np.abs(gp.K - Kref)))
Lref = np.linalg.cholesky(Kref + np.diag(fx["noise_variances"]))
out["L_err"] = float(np.max(np.abs(gp.Chol_factor - Lref)) / np.max(np.abs(Lref)))
import fvgp_amd.dist
fvgp_amd.dist.DEFAULT_OPS_FACTORY = StubOps          # the unpickled object builds its ops itself
gp2 = pickle.loads(pickle.dumps(gp))
out["pickle_loglik"] = gp2.log_likelihood()
out["pickle_pm"] = np.asarray(gp2.posterior_mean(fx["x_pred"])["m(x)"]).tolist()
del gp2
gp.set_hyperparameters(fx["thetas"][1])
out["loglik_set"] = gp.log_likelihood()
trained = gp.train(hyperparameter_bounds=np.array([[0.1, 10.0]] + [[0.05, 5.0]] * (len(fx["theta"]) - 1)), method="mcmc", max_iter=30)
out["trained"] = trained.tolist()
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


WORKER_SLOW = r'''
import os, sys, json, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
from dist_stub_ops import StubOps
from oracle import fvgp_oracle as orc
import fvgp_amd
dist.init_process_group(backend="gloo")
fx = load_golden("G2_rbf_n512_d3.npz")
x, y, th = fx["x"][:{n}], fx["y"][:{n}], fx["theta"]
warnings.simplefilter("ignore")
# (a) a Python kernel callable (4-argument form) in the row-sharded mode: every rank evaluates its own rows on the host
gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=fx["noise_variances"][:{n}], linalg_mode="CholInv",
                 kernel_function=lambda a, b, h, args: orc.rbf_ard(a, b, h), args={{"process_group": True, "shard_ops": StubOps(), "shard_panel": 128}})
out = dict(ll=gp.log_likelihood(), ll2=gp.log_likelihood(th * 1.05), pm=np.asarray(gp.posterior_mean(fx["x_pred"])["m(x)"]).tolist(),
           pS=gp.posterior_covariance(fx["x_pred"])["S"].tolist(), K_err=float(np.max(np.abs(gp.K - orc.rbf_ard(x, x, th)))))
# (b) a matrix-valued noise model: theta = [sig, l1, l2, l3, noise level]
def noise(xx, h):
    d2 = ((xx[:, None, :] - xx[None, :, :]) ** 2).sum(axis=2)
    return h[4] * (np.eye(len(xx)) + 0.3 * np.exp(-d2 / 0.02))
th5 = np.concatenate([th, [0.02]])
gpn = fvgp_amd.GP(x, y, init_hyperparameters=th5, kernel_function="rbf_ard", noise_function=noise,
                  args={{"process_group": True, "shard_ops": StubOps(), "shard_panel": 256}})
out["lln"] = gpn.log_likelihood()
out["pvn"] = gpn.posterior_covariance(fx["x_pred"], variance_only=True, add_noise=True)["v(x)"].tolist()
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close(); return port


def _run(world, n, d, panel, kernel, theta, ncol=1):
    code = WORKER.format(root=ROOT, n=n, d=d, panel=panel, kernel=kernel, theta=list(theta), ncol=ncol)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], "-c", code],
                         capture_output=True, text=True, env=env, timeout=600)
    if res.returncode != 0:
        # torchrun has no -c: fall back to a temp file
        raise RuntimeError(res.stdout[-2000:] + res.stderr[-4000:])
    import json
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def _run_file(world, tmp_path, template=None, **kw):
    code = (template or WORKER).format(root=ROOT, **kw)
    f = tmp_path / "worker.py"
    f.write_text(code)
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(f)],
                         capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("world,n,panel,kernel,ncol", [(2, 700, 256, "rbf_ard", 1), (3, 900, 128, "matern52_ard", 1),
                                                       (2, 1100, 384, "matern32_ard", 2), (4, 1300, 256, "rbf_ard", 1),
                                                       (4, 300, 128, "rbf_ard", 1), (8, 2100, 256, "rbf_ard", 1),
                                                       (5, 1500, 384, "matern52_ard", 2), (8, 2100, 1024, "rbf_ard", 1)])
def test_sharded_loglik_equals_dense(tmp_path, world, n, panel, kernel, ncol):
    d = 3
    theta = [1.0, 0.3, 0.35, 0.4]
    ll, logdet, quad = _run_file(world, tmp_path, n=n, d=d, panel=panel, kernel=kernel, theta=theta, ncol=ncol)
    x, y = synth(n, d)
    if ncol > 1:
        y = np.stack([y, np.cos(x.sum(axis=1))], axis=1)
    gp = orc.OracleGP(x, y, np.array(theta), np.full(n, 0.01), kernel=kernel)
    np.testing.assert_allclose(ll, gp.log_likelihood(), rtol=1e-10)
    np.testing.assert_allclose(logdet, gp.logdet_KV, rtol=1e-10)


@pytest.mark.parametrize("world,n,panel,kernel,ncol,npred", [(2, 700, 256, "rbf_ard", 1, 40), (3, 900, 128, "matern52_ard", 1, 130),
                                                             (4, 1000, 256, "matern32_ard", 2, 17), (3, 300, 384, "rbf_ard", 1, 5)])
def test_sharded_solves_posterior_gradient_equal_dense(tmp_path, world, n, panel, kernel, ncol, npred):
    """The distributed backward solve (KVinvY), the row-distributed forward solve behind the posterior covariance, and
    the gradient through each rank's partial Gram matrix of inv(L): all against the oracle, replicated on every rank
    (the reference's distributed == dense checks, tests/test_fvgp.py:3112-3183)."""
    d = 3
    theta = [1.1, 0.3, 0.35, 0.4]
    out = _run_file(world, tmp_path, template=WORKER_SOLVES, n=n, d=d, panel=panel, kernel=kernel, theta=theta, ncol=ncol, npred=npred)
    x, y = synth(n, d)
    if ncol > 1:
        y = np.stack([y, np.cos(x.sum(axis=1))], axis=1)
    gp = orc.OracleGP(x, y, np.array(theta), np.full(n, 0.01), kernel=kernel)
    assert out["replicated"]
    np.testing.assert_allclose(out["ll"], gp.log_likelihood(), rtol=1e-10)
    a = np.array(out["alpha"])
    assert np.max(np.abs(a - gp.KVinvY)) <= 1e-8 * np.max(np.abs(gp.KVinvY))
    xp = np.random.default_rng(7).random((npred, d))
    pm = gp.posterior_mean(xp)["m(x)_flat"].reshape(npred, -1) - np.mean(y)
    np.testing.assert_allclose(np.array(out["mean"]), pm, rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(np.array(out["S"]) - gp.posterior_covariance(xp)["S"])) <= 1e-10 * theta[0]
    g_ref = gp.neg_log_likelihood_gradient(np.array(theta), component=ncol - 1)
    np.testing.assert_allclose(np.array(out["g"]), g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))


@pytest.mark.parametrize("world,fixture,panel", [(2, "G2_rbf_n512_d3.npz", 256), (3, "G3_matern52_n512_d3.npz", 128)])
def test_gp_facade_routes_through_the_process_group(tmp_path, world, fixture, panel):
    """fvgp_amd.GP(..., args={"process_group": ...}): the reference's constructor-flag switch to distributed mode
    (gp.py:419-439).  Same public methods, compared with the reference's own golden outputs."""
    from conftest import load_golden
    out = _run_file(world, tmp_path, template=WORKER_FACADE, fixture=fixture, panel=panel)
    fx = load_golden(fixture)
    np.testing.assert_allclose(out["loglik"], fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(out["loglik_again"], fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(out["logliks"], fx["logliks"], rtol=1e-10)
    assert np.max(np.abs(np.array(out["KVinvY"]) - fx["KVinvY"])) <= 1e-8 * np.max(np.abs(fx["KVinvY"]))
    for key in ("grad", "grad_cached"):
        np.testing.assert_allclose(out[key], fx[key], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))
    np.testing.assert_allclose(out["pm"], fx["pm"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out["pm_theta1"], fx["pm_theta1"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(np.array(out["pS"]) - fx["pS"])) <= 1e-10 * fx["theta"][0] + 1e-12
    assert np.max(np.abs(np.array(out["pv_noise"]) - fx["pv_noise"])) <= 1e-10 * fx["theta"][0] + 1e-12
    assert out["K_err"] <= 1e-14 * fx["theta"][0] and out["L_err"] <= 1e-11
    np.testing.assert_allclose(out["pickle_loglik"], fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(out["pickle_pm"], fx["pm"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out["loglik_set"], fx["logliks"][1], rtol=1e-10)
    assert len(out["trained"]) == len(fx["theta"])


def test_single_rank_stub_path():
    """P = 1 without a process group (the same code path bench.py takes on one GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_stub_ops import StubOps
    from fvgp_amd.dist import ShardedGP
    n = 500
    x, y = synth(n, 2)
    gp = ShardedGP(x, y, np.full(n, 0.02), kernel="rbf_ard", ops=StubOps(), panel=256, rank=0, world=1)
    theta = np.array([1.2, 0.3, 0.5])
    ll, logdet, quad = gp.log_likelihood(theta)
    ref = orc.OracleGP(x, y, theta, np.full(n, 0.02), kernel="rbf_ard")
    np.testing.assert_allclose(ll, ref.log_likelihood(), rtol=1e-11)
    # block ownership bookkeeping
    assert gp.nblk == 4 and gp.nb_loc == 4 and gp.nv == n


def test_sharded_non_positive_definite_raises():
    """info travels on the device until the single read-back; the failing minor is reported in global rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_stub_ops import StubOps
    from fvgp_amd.dist import ShardedGP
    n = 600
    x, y = synth(n, 2)
    nv = np.full(n, 0.02); nv[300:] = -3.0
    gp = ShardedGP(x, y, nv, kernel="rbf_ard", ops=StubOps(), panel=256, rank=0, world=1)
    with pytest.raises(np.linalg.LinAlgError, match="leading minor"):
        gp.log_likelihood(np.array([1.0, 0.3, 0.5]))


def test_sharded_mode_takes_kernel_callables_and_matrix_valued_noise(tmp_path):
    """The reference's distributed mode answers the whole API (tests/test_fvgp.py:3112-3149): a Python kernel callable
    (each rank evaluates its own rows, gp_prior.py:217-224), linalg_mode="CholInv" (same results as "Chol"), the gathered K, and a
    2-d noise model (K + V, gp_kv.py:654-657; add_noise with the matrix, gp_posterior.py:554-569) through 3 ranks."""
    from conftest import load_golden
    n = 400
    out = _run_file(3, tmp_path, template=WORKER_SLOW, n=n)
    fx = load_golden("G2_rbf_n512_d3.npz")
    x, y, th, nv = fx["x"][:n], fx["y"][:n], fx["theta"], fx["noise_variances"][:n]
    ref = orc.OracleGP(x, y, th, nv, kernel="rbf_ard")
    np.testing.assert_allclose(out["ll"], ref.log_likelihood(), rtol=1e-10)
    np.testing.assert_allclose(out["ll2"], ref.log_likelihood(th * 1.05), rtol=1e-10)
    np.testing.assert_allclose(out["pm"], ref.posterior_mean(fx["x_pred"])["m(x)"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(np.array(out["pS"]) - ref.posterior_covariance(fx["x_pred"])["S"])) <= 1e-10
    assert out["K_err"] == 0.0
    # matrix-valued noise against plain numpy
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(axis=2)
    V = 0.02 * (np.eye(n) + 0.3 * np.exp(-d2 / 0.02))
    K = orc.rbf_ard(x, x, th)
    L = np.linalg.cholesky(K + V)
    ym = y - np.mean(y)
    a = np.linalg.solve(L.T, np.linalg.solve(L, ym))
    ll = -0.5 * (ym @ a + 2 * np.sum(np.log(np.diag(L))) + n * np.log(2 * np.pi))
    np.testing.assert_allclose(out["lln"], ll, rtol=1e-10)
    xp = fx["x_pred"]
    k = orc.rbf_ard(x, xp, th)
    v = np.linalg.solve(L, k)
    d2p = ((xp[:, None, :] - xp[None, :, :]) ** 2).sum(axis=2)
    Vp = 0.02 * (np.eye(len(xp)) + 0.3 * np.exp(-d2p / 0.02))
    var = np.diag(orc.rbf_ard(xp, xp, th) - v.T @ v) + np.diag(Vp)
    np.testing.assert_allclose(out["pvn"], var, rtol=1e-8, atol=1e-10)
