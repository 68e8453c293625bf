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


def _run_file(world, tmp_path, **kw):
    code = WORKER.format(root=ROOT, **kw)
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
