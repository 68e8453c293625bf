"""Row-sharded path with the real HIP ops: one rank (no process group) and two ranks sharing the one GPU of
the test box over gloo (RCCL refuses two ranks on one device; the collectives' semantics are the same)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth
from oracle import fvgp_oracle as orc

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
os.environ["FVGP_DEVICE"] = "0"          # every rank on the one GPU of the test box
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from fvgp_amd.dist import ShardedGP
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
n, d = {n}, 3
x, y = synth(n, d)
gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel={panel})
out = [gp.log_likelihood(np.array([1.0, 0.3, 0.3, 0.3]) * (1 + 0.02 * t)) for t in range(2)]
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


WORKER_FACADE = r'''
import os, sys, json, warnings
os.environ["FVGP_DEVICE"] = "0"          # every rank on the one GPU of the test box
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
import fvgp_amd
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
fx = load_golden({fixture!r})
warnings.simplefilter("ignore")
gp = fvgp_amd.GP(fx["x"], fx["y"], init_hyperparameters=fx["theta"], noise_variances=fx["noise_variances"],
                 kernel_function=str(fx["kernel"]), args={{"process_group": True, "shard_panel": {panel}}})
out = dict(loglik=gp.log_likelihood(), logliks=[gp.log_likelihood(t) for t in fx["thetas"]],
           KVinvY=gp.KVinvY.tolist(), grad=gp.neg_log_likelihood_gradient(fx["theta"]).tolist(),
           pm=np.asarray(gp.posterior_mean(fx["x_pred"])["m(x)"]).tolist(),
           pS=gp.posterior_covariance(fx["x_pred"])["S"].tolist())
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _spawn(tmp_path, code, world):
    """ranks started directly (no torchrun launcher process): GPU holders = pytest + `world` ranks <= 5"""
    f = tmp_path / "worker.py"
    f.write_text(code)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world))
        so, se = open(tmp_path / f"out{r}.txt", "w+"), open(tmp_path / f"err{r}.txt", "w+")
        procs.append((subprocess.Popen([sys.executable, str(f)], stdout=so, stderr=se, text=True, env=env), so, se))
    try:
        for p, _, _ in procs:
            p.wait(timeout=600)
    finally:
        for p, _, _ in procs:
            if p.poll() is None:
                p.kill()
    texts = []
    for p, so, se in procs:
        so.seek(0); se.seek(0)
        texts.append((so.read(), se.read()))
        so.close(); se.close()
        assert p.returncode == 0, texts[-1][0][-2000:] + texts[-1][1][-4000:]
    import json
    return json.loads([l for l in texts[0][0].splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("world,fixture,panel", [(2, "G2_rbf_n512_d3.npz", 256), (3, "G3_matern52_n512_d3.npz", 128)])
def test_gp_facade_over_a_process_group_with_the_hip_ops(tmp_path, world, fixture, panel):
    """GP(..., args={"process_group": True}) with the real HIP kernels under every rank (ranks share the one GPU of the
    box over gloo): log-likelihood, KVinvY from the distributed backward solve, gradient through the partial Gram
    matrices, posterior mean / covariance -- against the reference's golden outputs."""
    from conftest import load_golden
    out = _spawn(tmp_path, WORKER_FACADE.format(root=ROOT, fixture=fixture, panel=panel), world)
    fx = load_golden(fixture)
    np.testing.assert_allclose(out["loglik"], fx["loglik"], rtol=1e-10)
    np.testing.assert_allclose(out["logliks"], fx["logliks"], rtol=1e-10)
    assert np.max(np.abs(np.array(out["KVinvY"]) - fx["KVinvY"])) <= 1e-8 * np.max(np.abs(fx["KVinvY"]))
    np.testing.assert_allclose(out["grad"], fx["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))
    np.testing.assert_allclose(out["pm"], fx["pm"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(np.array(out["pS"]) - fx["pS"])) <= 1e-10 * fx["theta"][0] + 1e-12


def test_single_rank_sharded_solves_with_the_hip_ops():
    """One rank, no process group: the panel-wise backward solve, the row-distributed forward solve and the
    Gram-matrix gradient on the HIP kernels at a size with several panels, against the oracle."""
    from fvgp_amd.dist import ShardedGP
    n = 2500
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    gp = ShardedGP(x, y, nv, kernel="matern52_ard", panel=512, rank=0, world=1)
    ll, _, _ = gp.evaluate(theta, want_alpha=True)
    ref = orc.OracleGP(x, y, theta, nv, kernel="matern52_ard")
    np.testing.assert_allclose(ll, ref.log_likelihood(), rtol=1e-10)
    a = gp.alpha[:n, :1].cpu().numpy()
    assert np.max(np.abs(a - ref.KVinvY)) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    xp = np.random.default_rng(3).random((200, 3))
    mean, S = gp.posterior(xp)
    np.testing.assert_allclose(mean[:, 0] + np.mean(y), ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(S - ref.posterior_covariance(xp)["S"])) <= 1e-10
    g = gp.gradient()
    g_ref = ref.neg_log_likelihood_gradient(theta)
    np.testing.assert_allclose(g, g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))


def test_single_rank_hip_ops_match_fused_path():
    from fvgp_amd import _lib
    from fvgp_amd.dist import ShardedGP
    from fvgp_amd.device import default_handle
    n = 3000
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    gp = ShardedGP(x, y, nv, kernel="rbf_ard", panel=512, rank=0, world=1)
    ll, logdet, quad = gp.log_likelihood(theta)
    H = default_handle()
    npad = _lib.pad128(n)
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    ref = H.loglik(0, H.to_device(x), theta, H.to_device(nv), H.to_device((y - y.mean()).reshape(n, 1)), KV, alpha)
    np.testing.assert_allclose(ll, ref[0], rtol=1e-11)
    np.testing.assert_allclose(logdet, ref[1], rtol=1e-11)
    oracle, _ = orc.log_likelihood_once(x, y, nv, theta, "rbf_ard")
    np.testing.assert_allclose(ll, oracle, rtol=1e-10)


# The box allows at most 6 processes on its GPU: pytest + at most 4 ranks here; the 5- and 8-rank
# partitions are covered over gloo on the CPU (tests/test_dist_cpu.py).
@pytest.mark.parametrize("world,n,panel", [(2, 2000, 256), (3, 1700, 384), (4, 2500, 256)])
def test_ranks_sharing_one_gpu_over_gloo(tmp_path, world, n, panel):
    out = _spawn(tmp_path, WORKER.format(root=ROOT, n=n, panel=panel), world)
    x, y = synth(n, 3)
    for t, (ll, logdet, quad) in enumerate(out):
        ref, _ = orc.log_likelihood_once(x, y, np.full(n, 0.01), np.array([1.0, 0.3, 0.3, 0.3]) * (1 + 0.02 * t), "rbf_ard")
        np.testing.assert_allclose(ll, ref, rtol=1e-10)


WORKER_RCCL = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from fvgp_amd.dist import ShardedGP
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
n, d = {n}, 3
x, y = synth(n, d)
theta = np.array([1.0, 0.3, 0.3, 0.3])
gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel={panel}, force_collectives=True)
assert gp.collectives == "rccl" and gp.general
gp.ops.set_option("profile", 1)
ll = [gp.log_likelihood(theta * (1 + 0.02 * t))[0] for t in range(2)]
prof = gp.collective_summary()
ev = gp.evaluate(theta, want_alpha=True)
xp = np.random.default_rng(3).random((150, d))
mean, S = gp.posterior(xp)
g = gp.gradient()
t = torch.ones(4, dtype=torch.float64, device="cuda:0")
dist.all_reduce(t)                               # torch's own RCCL communicator next to the library's
print("RESULT " + json.dumps(dict(ll=ll, prof=prof, ev=list(ev), alpha=gp.alpha[:n, :1].cpu().numpy().ravel().tolist(),
                                  mean=mean.ravel().tolist(), S=S.tolist(), g=g.tolist(), t=float(t.sum().item()))))
dist.destroy_process_group()
'''


def test_one_rank_through_rccl(tmp_path):
    """The whole multi-rank code path on ONE GPU with the collectives really issued through RCCL (ncclCommInitRank on one
    rank, ncclAllGather / ncclAllReduce on the chain / main stream): ShardedGP(force_collectives=True) under
    backend="nccl".  Panel buffers, gather of the diagonal block, all-gather of the panel factor, replicated diagonal
    blocks in the solves -- against the oracle."""
    n, panel = 2500, 512
    out = _spawn(tmp_path, WORKER_RCCL.format(root=ROOT, n=n, panel=panel), 1)
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    for t, ll in enumerate(out["ll"]):
        ref, _ = orc.log_likelihood_once(x, y, nv, theta * (1 + 0.02 * t), "rbf_ard")
        np.testing.assert_allclose(ll, ref, rtol=1e-10)
    calls, nbytes, ms = out["prof"]["all_gather"]
    assert calls == 2 * (2 * 5 - 1) and ms > 0.0               # per evaluation: 5 diagonal blocks + 4 panel factors
    ref = orc.OracleGP(x, y, theta, nv, kernel="rbf_ard")
    np.testing.assert_allclose(out["ev"][0], ref.log_likelihood(), rtol=1e-10)
    assert np.max(np.abs(np.array(out["alpha"]) - ref.KVinvY[:, 0])) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    xp = np.random.default_rng(3).random((150, 3))
    np.testing.assert_allclose(np.array(out["mean"]) + np.mean(y), ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(np.array(out["S"]) - ref.posterior_covariance(xp)["S"])) <= 1e-10
    g_ref = ref.neg_log_likelihood_gradient(theta)
    np.testing.assert_allclose(out["g"], g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))
    assert out["t"] == 4.0


def test_forced_general_path_matches_the_in_place_path():
    """One rank without a process group: force_collectives routes through the panel buffers with copy collectives; the
    result is the in-place path's."""
    from fvgp_amd.dist import ShardedGP
    n = 3000
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    a = ShardedGP(x, y, nv, kernel="rbf_ard", panel=512, rank=0, world=1)
    b = ShardedGP(x, y, nv, kernel="rbf_ard", panel=512, rank=0, world=1, force_collectives=True, collectives="torch")
    la, lb = a.log_likelihood(theta), b.log_likelihood(theta)
    np.testing.assert_allclose(la, lb, rtol=1e-12)


def test_sharded_noise_function_gradient_uses_the_distributed_diagonal():
    """Gradients of noise-function hyperparameters in the row-sharded mode (diag(KV^-1) from the ranks' rows of inv(L),
    gp_marginal_likelihood.py:262-267) against the single-GPU facade."""
    import warnings
    import fvgp_amd
    n = 1500
    x, y = synth(n, 3)
    th = np.array([1.0, 0.3, 0.3, 0.3, 0.02])
    noise = lambda xx, h: np.full(len(xx), h[4]) * (1.0 + xx[:, 0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = fvgp_amd.GP(x, y, init_hyperparameters=th, kernel_function="rbf_ard", noise_function=noise)
        gp = fvgp_amd.GP(x, y, init_hyperparameters=th, kernel_function="rbf_ard", noise_function=noise,
                         args={"process_group": True, "shard_panel": 256, "shard_rank": 0, "shard_world": 1})
    g_ref, g = ref.neg_log_likelihood_gradient(th), gp.neg_log_likelihood_gradient(th)
    np.testing.assert_allclose(g, g_ref, rtol=1e-7, atol=1e-8 * np.max(np.abs(g_ref)))


def test_bench_multi_rank_line_is_the_sharded_evaluation(tmp_path):
    """bench.py with 2 ranks (sharing the one GPU over gloo): ONE JSON line whose headline is the row-sharded evaluation
    ("scaling": "strong"), with the collectives' traffic, the agreement with a single-GPU evaluation of the same theta
    and the replicas side record."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, FVGP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--npoints", "12000", "--backend", "gloo"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "cpu_baseline" not in out and "error" not in out
    assert out["rel_diff_vs_single_gpu"] < 1e-10
    assert out["collectives"]["all_gather"]["bytes_received_per_rank_per_eval"] > 0
    assert out["roofline"]["launches"] > 0 and 0.0 < out["roofline"]["frac"] < 1.0
    assert out["replicas"]["value"] > 0 and out["replicas"]["scaling"] == "weak"


def test_bench_plain_python_form_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher around it (the form the driver uses): bench.py itself starts the two ranks as
    children of a parent that never touches the GPU, relays rank 0's ONE JSON line and its exit code.  The line says what the
    communicator saw (2 ranks), carries the A/B of the two collective backends (the process group's own and the direct IPC pulls, two
    evaluations each) with the faster one timed, per-collective milliseconds and per-link rates, and agrees with one GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FVGP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--npoints", "12000", "--backend", "gloo"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "error" not in out
    assert out["ranks_seen_by_communicator"] == 2 and out["communicator"]["torch_distributed"]["world_size"] == 2
    ab = out["collectives_ab"]
    assert set(ab) == {"torch", "ipc"} and all("ms_per_eval" in r and "error" not in r for r in ab.values()), ab
    assert abs(ab["torch"]["loglik"] - ab["ipc"]["loglik"]) <= 1e-10 * abs(ab["ipc"]["loglik"])
    assert out["collectives_chosen"] == min(ab, key=lambda k: ab[k]["ms_per_eval"])
    assert out["rel_diff_vs_single_gpu"] < 1e-10
    ag = out["collectives"]["all_gather"]
    assert ag["bytes_received_per_rank_per_eval"] > 0 and ag["ms_per_call"] > 0 and ag["GBps_per_peer_link"] > 0
    assert out["roofline"]["launches"] > 0 and 0.0 < out["roofline"]["frac"] < 1.0


def test_bench_keeps_the_completed_backend_when_the_other_hangs(tmp_path):
    """--collectives auto runs the process group's own collectives through the whole timed region FIRST; if the direct pulls tried after
    them never return (test hook), the watchdog prints the completed backend's line -- a full measurement, not value 0.0 -- with the
    hang recorded under collectives_ab, and the run exits 0: one lease on a multi-GPU node is not lost to the experimental backend."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FVGP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--npoints", "4000", "--backend", "gloo", "--inject-ipc-hang", "--sharded-timeout", "45"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == 2 and out["collectives_chosen"] == "torch" and out["steps"] == 2
    assert "ms_per_eval" in out["collectives_ab"]["torch"] and "did not complete" in out["collectives_ab"]["ipc"]["error"]
    assert out["rel_diff_vs_single_gpu"] < 1e-10


def test_ipc_collective_that_never_completes_fails_loudly(tmp_path):
    """A peer that never shows up: the poll of the direct collectives gives up (FVGP_IPC_TIMEOUT_S), the copies behind it have run
    on stale windows -- and the synchronisation that follows must FAIL with status 2200 (it used to return 0 and hand the
    stale result to the host; only the NEXT collective noticed)."""
    code = r'''
import os, sys, json
os.environ["FVGP_DEVICE"] = "0"; os.environ["FVGP_IPC_TIMEOUT_S"] = "2"
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from fvgp_amd import _lib
from fvgp_amd.dist import HipOps
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
o = HipOps(_lib.Handle(0))
both = [None, None]
dist.all_gather_object(both, o.ipc_window(1 << 20))
name = [f"/fvgp_ipc_test_{{os.getpid()}}" if rank == 0 else None]
dist.broadcast_object_list(name, src=0)
o.comm_init_ipc(both, name[0], rank, 2)
dist.barrier()
if rank == 0:
    os.unlink("/dev/shm" + name[0])
    send, recv = o.zeros(1024) + 1.0, o.zeros(2048)
    o.all_gather(send, recv)                   # rank 1 never calls its side
    try:
        o.sync()
        print('RESULT "no error"')
    except _lib.HipExtensionError as e:
        print("RESULT " + json.dumps("2200" if "2200" in str(e) else str(e)))
dist.barrier()
o.close()
dist.destroy_process_group()
'''
    out = _spawn(tmp_path, code.format(root=ROOT), 2)
    assert out == "2200", out


def test_bench_line_when_the_sharded_evaluation_raises(tmp_path):
    """bench.py with 2 ranks whose sharded evaluation raises (test hook): the ONE line says value 0.0 + error like the watchdog's,
    the replicas ride along as a side record, the exit code is 4."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, FVGP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--npoints", "4000", "--backend", "gloo",
                          "--inject-sharded-failure"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert res.returncode != 0
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads(lines[0])
    assert out["value"] == 0.0 and out["scaling"] == "strong" and "injected failure" in out["error"]
    assert out["replicas"]["value"] > 0 and out["replicas"]["scaling"] == "weak"


WORKER_FULL = r'''
import os, sys, json
os.environ["FVGP_DEVICE"] = "0"          # every rank on the one GPU of the test box
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from fvgp_amd.dist import ShardedGP
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
n, d = {n}, 3
x, y = synth(n, d)
theta = np.array([1.0, 0.3, 0.3, 0.3])
gp = ShardedGP(x, y, np.full(n, 0.01), kernel={kernel!r}, panel={panel})
lls = [gp.log_likelihood(theta * (1 + 0.02 * t))[0] for t in range(2)]
ev = gp.evaluate(theta, want_alpha=True)
xp = np.random.default_rng(3).random(({npred}, d))
mean, S = gp.posterior(xp)
g = gp.gradient()
if dist.get_rank() == 0:
    np.savez({out!r}, lls=np.array(lls), ev=np.array(ev), alpha=gp.alpha[:n, 0].cpu().numpy(), mean=mean, S=S, g=g)
    print("RESULT " + json.dumps(dict(ok=True)))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,n,panel,kernel,npred", [
    (4, 3000, 512, "rbf_ard", 150),            # one 128-block per rank and panel: the diagonal block is gathered IN PLACE (dist_driver.h, chain step 1)
    (2, 20000, 1024, "rbf_ard", 200),          # C2's size at the default panel width
    (3, 9000, 2048, "rbf_ard", 130),           # the width bench.py takes from N = 40 000 on (half as many collectives), three ranks
])
def test_default_panel_shapes_over_gloo_against_the_oracle(tmp_path, world, n, panel, kernel, npred):
    """The row-sharded path at the panel shapes an 8-GPU run takes by default -- 1024-wide panels, and the in-place gather of the
    diagonal block (ranks == blocks per panel) -- on the HIP ops with more than one rank (ranks share the box's GPU, gloo
    callbacks): log-likelihood, KVinvY, posterior mean / covariance and gradient against the oracle on the same inputs."""
    out = str(tmp_path / "full.npz")
    _spawn(tmp_path, WORKER_FULL.format(root=ROOT, n=n, panel=panel, kernel=kernel, npred=npred, out=out), world)
    r = np.load(out)
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    for t, ll in enumerate(r["lls"]):
        ref, _ = orc.log_likelihood_once(x, y, nv, theta * (1 + 0.02 * t), kernel)
        np.testing.assert_allclose(ll, ref, rtol=1e-10)
    ref = orc.OracleGP(x, y, theta, nv, kernel=kernel)
    np.testing.assert_allclose(r["ev"][0], ref.log_likelihood(), rtol=1e-10)
    np.testing.assert_allclose(r["ev"][1], ref.logdet_KV, rtol=1e-10)
    assert np.max(np.abs(r["alpha"] - ref.KVinvY[:, 0])) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    xp = np.random.default_rng(3).random((npred, 3))
    np.testing.assert_allclose(r["mean"][:, 0] + np.mean(y), ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(r["S"] - ref.posterior_covariance(xp)["S"])) <= 1e-10
    g_ref = ref.neg_log_likelihood_gradient_potri(theta) if n > 4000 else ref.neg_log_likelihood_gradient(theta)
    np.testing.assert_allclose(r["g"], g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))


WORKER_IPC = r'''
import os, sys, json
os.environ["FVGP_DEVICE"] = "0"          # every rank on the one GPU of the test box
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import synth
from fvgp_amd.dist import ShardedGP
torch.cuda.set_device(0)
if {world} > 1:
    dist.init_process_group(backend="gloo")          # the bootstrap only: window handles and the flag file's name
n, d = {n}, 3
x, y = synth(n, d)
theta = np.array([1.0, 0.3, 0.3, 0.3])
gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel={panel}, collectives="ipc", force_collectives=True,
               ipc_window_bytes={window}, **({{}} if {world} > 1 else dict(rank=0, world=1)))
assert gp.collectives == "ipc" and gp.general
gp.ops.set_option("profile", 1)
lls = [gp.log_likelihood(theta * (1 + 0.02 * t))[0] for t in range(3)]
prof = gp.collective_summary()
ev = gp.evaluate(theta, want_alpha=True)
xp = np.random.default_rng(3).random((150, d))
mean, S = gp.posterior(xp)
g = gp.gradient(slab=512)
rank = dist.get_rank() if {world} > 1 else 0
if rank == 0:
    np.savez({out!r}, lls=np.array(lls), ev=np.array(ev), alpha=gp.alpha[:n, 0].cpu().numpy(), mean=mean, S=S, g=g,
             calls=np.array([prof["all_gather"][0], prof["all_reduce"][0]]), ms=np.array([prof["all_gather"][2], prof["all_reduce"][2]]))
    print("RESULT " + json.dumps(dict(ok=True)))
gp.close()
if {world} > 1:
    dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,n,panel,window", [(1, 2500, 512, 0), (2, 6000, 1024, 0), (4, 3000, 512, 0), (3, 2500, 256, 64 * 1024)])
def test_direct_ipc_collectives_against_the_oracle(tmp_path, world, n, panel, window):
    """The row-sharded path with the direct collectives of csrc/ipc.hip (collectives="ipc": payload by hipMemcpyAsync between IPC
    mappings of the ranks' windows, flags in a shared-memory file, one-wave poll / flag kernels) instead of a collective library:
    1 - 4 ranks sharing the box's GPU, gloo only as the bootstrap.  Likelihood, KVinvY, posterior and gradient against the oracle;
    the last case has a window so small that every large call is cut into many pieces."""
    out = str(tmp_path / "ipc.npz")
    _spawn(tmp_path, WORKER_IPC.format(root=ROOT, n=n, panel=panel, world=world, window=window or None, out=out), world)
    r = np.load(out)
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    for t, ll in enumerate(r["lls"]):
        ref, _ = orc.log_likelihood_once(x, y, nv, theta * (1 + 0.02 * t), "rbf_ard")
        np.testing.assert_allclose(ll, ref, rtol=1e-10)
    assert r["calls"][0] > 0 and r["ms"][0] > 0.0
    ref = orc.OracleGP(x, y, theta, nv, kernel="rbf_ard")
    np.testing.assert_allclose(r["ev"][0], ref.log_likelihood(), rtol=1e-10)
    assert np.max(np.abs(r["alpha"] - ref.KVinvY[:, 0])) <= 1e-8 * np.max(np.abs(ref.KVinvY))
    xp = np.random.default_rng(3).random((150, 3))
    np.testing.assert_allclose(r["mean"][:, 0] + np.mean(y), ref.posterior_mean(xp)["m(x)"], rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(r["S"] - ref.posterior_covariance(xp)["S"])) <= 1e-10
    g_ref = ref.neg_log_likelihood_gradient_potri(theta) if n > 4000 else ref.neg_log_likelihood_gradient(theta)
    np.testing.assert_allclose(r["g"], g_ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(g_ref)))
