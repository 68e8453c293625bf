"""CPU-only checks: the C-ABI library builds, loads and exports every symbol the header declares;
host-side logic (tile map, index-set transform, MCMC driver, argument rules) -- no GPU compute."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import fvgp_oracle as orc


@pytest.fixture(scope="module")
def L():
    from fvgp_amd import _lib
    _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from fvgp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fvgp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fvgp_hip_\w+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == declared
    for s in declared:
        assert hasattr(L, s), f"libfvgp_hip.so does not export {s}"
    assert L.fvgp_hip_version() >= 100
    assert L.fvgp_hip_padded_dim(1) == 128 and L.fvgp_hip_padded_dim(128) == 128 and L.fvgp_hip_padded_dim(50000) == 50048


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from fvgp_amd import _lib
    with pytest.raises(_lib.HipExtensionError):
        _lib.Handle(0)
    import fvgp_amd
    x = np.random.default_rng(0).random((20, 2)); y = np.sin(x[:, 0])
    with pytest.raises(_lib.HipExtensionError):
        fvgp_amd.GP(x, y, init_hyperparameters=np.ones(3), noise_variances=np.full(20, 0.01))


def test_cpu_device_is_refused():
    """tests/test_fvgp.py:4682-4702 analogue: an unknown/unsupported compute device is an exception, never
    a silent fallback."""
    import fvgp_amd
    x = np.random.default_rng(0).random((20, 2)); y = np.sin(x[:, 0])
    with pytest.raises(Exception, match="No valid compute device"):
        fvgp_amd.GP(x, y, init_hyperparameters=np.ones(3), noise_variances=np.full(20, 0.01), compute_device="cpu")
    with pytest.raises(NotImplementedError):
        fvgp_amd.GP(x, y, init_hyperparameters=np.ones(3), noise_variances=np.full(20, 0.01), gp2Scale=True)
    with pytest.raises(Exception, match="Decide which one"):
        fvgp_amd.GP(x, y, init_hyperparameters=np.ones(3), noise_variances=np.full(20, 0.01),
                    noise_function=lambda x, h: np.ones(len(x)))


@pytest.mark.parametrize("case", [(1, 1, 0), (3, 1, 0), (391, 1, 0), (5, 3, 1), (12, 3, 1), (8, 8, 1), (9, 9, 1),
                                         (23, 23, 1), (40, 16, 1), (17, 5, 0), (16, 24, 0), (391, 391, 1), (100, 7, 1),
                                         # row-sharded predicate tj <= ti*scale + off (lower = 2)
                                         (49, 383, 2, 8, 3), (50, 391, 2, 8, -5), (391, 383, 2, 1, 0), (13, 8, 2, 2, -9),
                                         (3, 5, 2, 4, 0), (20, 100, 2, 3, -70)])
def test_gemm_tile_map_covers_each_tile_once(L, case):
    """The blockIdx -> tile map (8xSN super-tiles + XCD remap) must hit every wanted tile exactly once."""
    tm, tn, lower, scale, off = (tuple(case) + (1, 0))[:5]
    cap = 400000
    ti = (ctypes.c_int * cap)(); tj = (ctypes.c_int * cap)()
    nwg = L.fvgp_hip_debug_tile_map(tm, tn, lower, scale, off, ti, tj, cap)
    assert 0 <= nwg <= cap
    keep = (lambda i, j: True) if lower == 0 else (lambda i, j: j <= i) if lower == 1 else (lambda i, j: j <= i * scale + off)
    got = {}
    for b in range(nwg):
        i, j = ti[b], tj[b]
        if i >= tm or j >= tn or not keep(i, j):
            continue
        assert (i, j) not in got, f"tile {(i, j)} mapped twice"
        got[(i, j)] = b
    want = {(i, j) for i in range(tm) for j in range(tn) if keep(i, j)}
    assert set(got) == want
    if lower == 2 and want:
        assert nwg <= 2 * len(want) + 64 * ((tm + 7) // 8)       # the grid carries no dead half
    # blocks b and b+8 share an XCD: their tiles should be neighbours in the enumeration
    if nwg >= 64 and tn >= 8 and lower != 2:
        i0, j0, i1, j1 = ti[0], tj[0], ti[8], tj[8]
        assert abs(i0 - i1) <= 8 and abs(j0 - j1) <= 8


@pytest.mark.parametrize("n,n2", [(1, 1), (1, 9), (4, 4), (4, 32), (8, 8), (8, 57), (16, 16), (16, 235), (32, 32), (32, 63), (32, 391)])
@pytest.mark.parametrize("ahead,slots", [(0, 1), (3, 16), (3, 512), (8, 40)])
def test_panel_kernel_ticket_order_makes_progress_and_bulk_tasks_wait_for_lower_tickets_only(L, n, n2, ahead, slots):
    """The resident panel kernel (csrc/chain.hip) deals its tasks by a start-order ticket and never needs all workgroups resident.  Host
    replay of the ticket -> task map (the function the kernel calls).  Per block column the diagonal block and the two blocks under it
    are critical, the rest bulk; the critical tasks of column k + ahead come right before the bulk tasks of column k:
      * every block of the first n2 block rows has exactly one ticket; later tickets are whole block rows;
      * everything a BULK task waits for -- the blocks left of it in its row and in the row of its column's diagonal block, that
        diagonal block's leaf -- has a LOWER ticket (ahead = 0: that holds for every task: plain column order);
      * with `slots` workgroup slots, a task holding its slot until everything it waits for has completed, the order never stalls
        (ahead = 3 with 16 slots: at most 12 critical tasks are ever ahead)."""
    out = (ctypes.c_int * 3)()
    nsq = n * n2 - n * (n - 1) // 2
    ticket, order = {}, []
    for t in range(nsq + 3):
        assert L.fvgp_hip_debug_chain_ticket(n, n2, ahead, t, out) == 0
        kind, row, col = out[0], out[1], out[2]
        if t < nsq:
            assert kind in (0, 1) and 0 <= col < n and col <= row < n2 and (kind == 0) == (row == col)
            assert (row, col) not in ticket
            ticket[(row, col)] = t
            order.append((row, col))
        else:
            assert kind == 2 and row == n2 + (t - nsq) and col == -1
    assert set(ticket) == {(r, k) for k in range(n) for r in range(k, n2)}

    def waits(row, k):
        if row == k:
            return [(k, j) for j in range(k)]                                  # the solved blocks of its row, column by column
        return [(row, j) for j in range(k)] + [(k, j) for j in range(k)] + [(k, k)]      # both operands of its products; the leaf

    for (row, k), t in ticket.items():
        if ahead == 0 or row - k > 2:
            for w in waits(row, k):
                assert ticket[w] < t, (n, n2, ahead, (row, k), "waits for", w)
        else:                                                                      # a critical task: at most `ahead` columns in front
            for w in waits(row, k):
                assert ticket[w] < t or (w[0] - w[1] > 2 and k - w[1] <= ahead), (n, n2, ahead, (row, k), "waits for", w)
    done, running, nxt = set(), [], 0
    while len(done) < nsq:
        while len(running) < slots and nxt < nsq:
            running.append(order[nxt]); nxt += 1
        finished = [t for t in running if all(w in done for w in waits(*t))]
        assert finished, (n, n2, ahead, slots, "stalled with", running[:6])
        done.update(finished)
        running = [t for t in running if t not in finished]


@pytest.mark.parametrize("case", [(391, 391, 1), (131, 131, 1), (100, 100, 1), (64, 64, 1), (375, 16, 1), (40, 300, 0),
                                         (47, 359, 2, 8, -8), (19, 135, 2, 8, -8), (9, 55, 2, 8, -8), (50, 391, 2, 8, -5), (3, 5, 2, 4, 0)])
def test_balanced_tile_table(L, case):
    """The XCD-balanced block -> tile table: every wanted tile exactly once, and the eight XCDs (blocks b, b + 8, ...) get the
    same number of real tiles to within one -- the formula map leaves up to 25 % more on the fullest XCD."""
    tm, tn, lower, scale, off = (tuple(case) + (1, 0))[:5]
    cap = 400000
    tab = (ctypes.c_int * cap)()
    grid = L.fvgp_hip_debug_tile_table(tm, tn, lower, scale, off, tab, cap)
    assert 0 <= grid <= cap and grid % 8 == 0
    keep = (lambda i, j: True) if lower == 0 else (lambda i, j: j <= i) if lower == 1 else (lambda i, j: j <= i * scale + off)
    want = {(i, j) for i in range(tm) for j in range(tn) if keep(i, j)}
    got, per_xcd = set(), [0] * 8
    for b in range(grid):
        e = tab[b]
        if e < 0:
            continue
        t = (e >> 16, e & 0xffff)
        assert t not in got
        got.add(t)
        per_xcd[b % 8] += 1
    assert got == want
    assert max(per_xcd) - min(per_xcd) <= 1
    assert grid == 8 * max(per_xcd) if want else grid == 0
    # an XCD walks its run in order: entries of one XCD are neighbours in the super-tile enumeration
    if len(want) >= 1024 and lower != 2:
        e0, e1 = tab[0], tab[8]
        assert abs((e0 >> 16) - (e1 >> 16)) <= 8 and abs((e0 & 0xffff) - (e1 & 0xffff)) <= 8


def test_index_set_transform_and_cartesian_product():
    from fvgp_amd.fvgp import transform_index_set
    from fvgp_amd.gp import GP
    for name in ("G5_fvgp_4x64.npz", "G5n_fvgp_4x64_nan.npz"):
        fx = load_golden(name)
        x, y, v = transform_index_set(fx["fvgp_x"], fx["fvgp_y"], fx["fvgp_noise"], 4)
        assert np.array_equal(x, fx["x"]) and np.array_equal(y, fx["y"]) and np.array_equal(v, fx["noise_variances"])
        assert np.array_equal(GP.cartesian_product(fx["x_pred"], fx["x_out"]), fx["pm_xpred"])
        assert np.array_equal(GP.cartesian_product(fx["x_pred"], fx["x_out"]), orc.cartesian_product(fx["x_pred"], fx["x_out"]))


def test_mcmc_driver_on_a_toy_posterior():
    """run_mcmc (gp_mcmc.py:96-224 restated) finds the mode of a Gaussian log-density in a box."""
    from fvgp_amd import gp_training
    mu = np.array([1.0, 2.0])
    f = lambda t: -0.5 * np.sum(((t - mu) / 0.1) ** 2)
    bounds = np.array([[0.0, 4.0], [0.0, 4.0]])
    res = gp_training.run_mcmc(f, bounds, np.array([3.0, 0.5]), n_updates=3000, rng=np.random.default_rng(3))
    assert np.all(np.abs(res["median(x)"] - mu) < 0.1)
    assert res["max f(x)"] > -0.5
    assert np.all(res["x"] >= bounds[:, 0]) and np.all(res["x"] <= bounds[:, 1])
    with pytest.raises(Exception, match="outside of optimization bounds"):            # gp_training.py:54-55
        gp_training.train(None, bounds, np.array([5.0, 1.0]))
    with pytest.raises(ValueError, match="No optimization mode"):                      # gp_training.py:195
        gp_training.train(None, bounds, np.array([1.0, 1.0]), method=42, objective_function=f)

    class Holder:
        pass
    # a callable method gets the object and must hand back a 1-d ndarray (gp_training.py:194-196)
    h = Holder()
    np.testing.assert_array_equal(gp_training.train(h, bounds, np.array([1.0, 1.0]), method=lambda g: mu, objective_function=f), mu)
    with pytest.raises(AssertionError, match="invalid hyperparameters"):
        gp_training.train(h, bounds, np.array([1.0, 1.0]), method=lambda g: list(mu), objective_function=f)
    # a log prior of -inf vetoes proposals without a likelihood call (gp_mcmc.py:203-209)
    seen = []

    def fcount(t):
        seen.append(t.copy())
        return f(t)
    res = gp_training.run_mcmc(fcount, bounds, np.array([0.5, 2.0]), n_updates=400, rng=np.random.default_rng(4),
                               prior=lambda t, box, args: 0.0 if (gp_training._in_bounds(t, box) and t[0] < 0.8) else -np.inf)
    assert np.all(res["x"][:, 0] < 0.8) and all(t[0] < 0.8 for t in seen)
    # a user objective with its gradient drives 'local' and 'adam'
    g = lambda t: (t - mu) / 0.01
    nf = lambda t: -f(t)
    np.testing.assert_allclose(gp_training.train(h, bounds, np.array([3.0, 0.5]), method="local", objective_function=nf,
                                                 objective_function_gradient=g, tolerance=1e-12), mu, atol=1e-6)
    got = gp_training.train(h, bounds, np.array([1.2, 1.8]), method="adam", objective_function=nf,
                            objective_function_gradient=g, max_iter=2000)
    assert np.all(np.abs(got - mu) < 0.05) and len(h.adam_history["nlml"]) >= 1


def test_training_drivers_walk_the_reference_traces():
    """gp_training.run_mcmc / adam_optimize around the ORACLE's likelihood reproduce the reference's own seeded chain and
    Adam history (G11: numpy's legacy global stream, gp_mcmc.py:214,337-356; gp_training.py:576-667) -- the host logic of
    the callers, without a GPU."""
    from conftest import load_golden
    from oracle import fvgp_oracle as orc
    from fvgp_amd import gp_training
    fx = load_golden("G11_training_traces_m52_n200_d2.npz")
    o = orc.OracleGP(fx["x"], fx["y"], fx["theta"], fx["noise_variances"], kernel="matern52_ard")
    res = gp_training.run_mcmc(o.log_likelihood, fx["bounds"], fx["theta"], n_updates=int(fx["mcmc_max_iter"]),
                               rng=np.random.RandomState(int(fx["seed"])))
    assert res["x"].shape == fx["mcmc_x"].shape
    np.testing.assert_allclose(res["x"], fx["mcmc_x"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(res["f(x)"], fx["mcmc_f"], rtol=1e-10)
    np.testing.assert_allclose(res["median(x)"], fx["mcmc_median"], rtol=1e-12)
    # the same through the global stream, as a user of the reference would seed it
    np.random.seed(int(fx["seed"]))
    res2 = gp_training.run_mcmc(o.log_likelihood, fx["bounds"], fx["theta"], n_updates=int(fx["mcmc_max_iter"]))
    np.testing.assert_array_equal(res2["x"], res["x"])
    th, hist = gp_training.adam_optimize(lambda t: -o.log_likelihood(t), o.neg_log_likelihood_gradient, fx["theta"],
                                         max_iter=int(fx["adam_max_iter"]))
    np.testing.assert_allclose(np.asarray(hist["theta"]), fx["adam_theta"], rtol=1e-7)
    np.testing.assert_allclose(np.asarray(hist["nlml"]), fx["adam_nlml"], rtol=1e-9)
    np.testing.assert_allclose(th, fx["adam_result"], rtol=1e-7)


def test_named_kernel_resolution():
    from fvgp_amd import kernels
    assert kernels.resolve(None) is kernels.matern32_ard          # gp_prior.py:62-63 default
    assert kernels.resolve("rbf_ard") is kernels.rbf_ard
    assert kernels.resolve(kernels.matern52_iso).kernel_id == 5
    assert kernels.resolve(lambda a, b, h: None) is None
    with pytest.raises(ValueError):
        kernels.resolve("nope")
    assert kernels.rbf_ard.n_hyperparameters(3) == 4 and kernels.rbf_iso.n_hyperparameters(3) == 2


def test_workspace_bytes_query(L):
    """Handle-scoped device memory is small and monotone; every N x N buffer is the caller's (SURVEY 8b ownership)."""
    w50 = L.fvgp_hip_workspace_bytes(50000, 0)
    assert 50e6 < w50 < 60e6                       # 391 inverted 128x128 blocks dominate
    assert L.fvgp_hip_workspace_bytes(50000, 1000) > w50
    assert L.fvgp_hip_workspace_bytes(20000, 0) < w50
    assert L.fvgp_hip_workspace_bytes(0, 0) == -1 and L.fvgp_hip_workspace_bytes(10, -1) == -1


def test_host_kernel_building_blocks_equal_the_oracles():
    """fvgp_amd.kernels' helpers for user-written host callables against the oracle's restatement
    of fvgp/kernels.py:16-33,98-118,166-188,440-481 -- same names, same values"""
    from fvgp_amd import kernels
    from oracle import fvgp_oracle as orc
    rng = np.random.default_rng(5)
    x1, x2 = rng.random((17, 3)), rng.random((9, 3))
    ell = np.array([0.3, 1.7, 0.9])
    np.testing.assert_allclose(kernels.get_distance_matrix(x1, x2), orc.get_distance_matrix(x1, x2), rtol=1e-14, atol=1e-15)
    d = kernels.get_anisotropic_distance_matrix(x1, x2, ell)
    np.testing.assert_allclose(d, orc.get_anisotropic_distance_matrix(x1, x2, ell), rtol=1e-14, atol=1e-15)
    assert d.shape == (17, 9)
    for name in ("squared_exponential_kernel", "matern_kernel_diff1", "matern_kernel_diff2"):
        for length in (1.0, 0.37):
            np.testing.assert_allclose(getattr(kernels, name)(d, length), getattr(orc, name)(d, length), rtol=1e-14)
    np.testing.assert_allclose(kernels.exponential_kernel(d, 0.5), np.exp(-d / 0.5), rtol=1e-15)


def test_user_proposal_distributions_drive_the_mcmc():
    """gp_mcmc.py:234-364: ProposalDistribution objects.  One normal proposal over all indices walks the default chain
    (same random stream, same adaptation); two objects -- a normal one and a user callable with its own adapt function --
    each move their own entries once per iteration."""
    from fvgp_amd import gp_training as T
    b = np.array([[0.1, 5.0], [0.05, 3.0]])
    f = lambda x: -0.5 * np.sum((x - np.array([1.0, 0.7])) ** 2 / 0.05)
    x0 = np.array([2.0, 1.0])
    np.random.seed(3)
    a = T.run_mcmc(f, b, x0, n_updates=300, break_default=False)
    np.random.seed(3)
    std = (b[:, 1] - b[:, 0]) * 0.2 / np.sqrt(12)
    c = T.run_mcmc_proposals(f, b, x0, [T.ProposalDistribution(np.arange(2), init_prop_Sigma=np.diag(std ** 2))],
                             n_updates=300, break_default=False)
    np.testing.assert_allclose(c["x"], a["x"], rtol=0, atol=1e-12)
    calls = []

    def shrink(end, chain):                              # user adapt: (iteration, chain) -> None, updates prop_args
        calls.append(len(chain.trace["x"]))
        pd2.prop_args["width"] *= 0.999

    def uniform_step(x_part, x_all, obj):                # user proposal: (own entries, all entries, obj) -> new own entries
        return x_part + obj.rng.uniform(-obj.prop_args["width"], obj.prop_args["width"], len(x_part))

    pd1 = T.ProposalDistribution([0], init_prop_Sigma=np.array([[0.05]]))
    pd2 = T.ProposalDistribution([1], proposal_dist=uniform_step, adapt_callable=shrink, prop_args={"width": 0.2})
    res = T.run_mcmc_proposals(f, b, x0, [pd1, pd2], n_updates=400, rng=np.random.RandomState(5), break_default=False)
    assert res["x"].shape == (400, 2) and len(calls) == 399 and len(pd1.jump_trace) == len(pd2.jump_trace) == 399
    assert 0.0 < res["acceptance"] < 1.0 and np.all((res["x"] >= b[:, 0]) & (res["x"] <= b[:, 1]))
    assert abs(res["mean(x)"][0] - 1.0) < 0.6 and abs(res["mean(x)"][1] - 0.7) < 0.6
    with pytest.raises(Exception, match="No proposal distribution"):
        T.ProposalDistribution([0], proposal_dist=3)


def test_isa_lint_of_the_built_code_objects():
    """tools/isa_lint.py over the objects the library was linked from (the Makefile runs the same lint after every link): no 16-byte
    buffer store with an SGPR offset (the gfx950 store hazard of profiles/r05_store_hazard_chain_verify.txt), a wait state behind
    every wide store, sc1 on every vector load of the resident panel kernel.  The rules themselves are exercised on hand-written
    disassembly lines, so a lint that stopped matching anything cannot pass silently."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import glob
    import isa_lint
    objs = sorted(glob.glob(os.path.join(ROOT, "fvgp_amd", "csrc", "*.o")))
    assert objs, "no objects: build() first"
    bad, seen = isa_lint.lint_objects(objs)
    assert seen >= 6 and not bad, bad[:5]
    head = "0000000000001000 <_ZN12_GLOBAL__N_112chain_kernelILb0EEEvNS_9ChainArgsE>:\n"
    r1 = head + "\tbuffer_store_dwordx4 v[2:5], v77, s[8:11], s70 offen   // 0: 0\n\ts_nop 0\n"
    r2 = head + "\tbuffer_store_dwordx4 v[2:5], v77, s[8:11], 0 offen offset:16 // 0: 0\n\tv_add_f64 v[4:5], v[10:11], v[12:13] // 0\n"
    ok2 = head + "\tbuffer_store_dwordx4 v[2:5], v77, s[8:11], 0 offen // 0: 0\n\ts_nop 0\n\tv_add_f64 v[4:5], v[10:11], v[12:13] // 0\n"
    g2 = head + "\tglobal_store_dwordx4 v[8:9], v[20:23], off sc1 // 0\n\tv_mov_b32_e32 v21, v3 // 0\n"
    r3 = head + "\tbuffer_load_dwordx4 v[2:5], v77, s[8:11], 0 offen // 0\n"
    ok3 = head + "\tbuffer_load_dwordx4 v[2:5], v77, s[8:11], 0 offen sc1 // 0\n\tbuffer_load_dword v1, s[4:7], 0 offen sc1 lds // 0\n"
    assert [b[0] for b in isa_lint.lint_text(r1, "t")] == ["R1"]
    assert [b[0] for b in isa_lint.lint_text(r2, "t")] == ["R2"]
    assert [b[0] for b in isa_lint.lint_text(g2, "t")] == ["R2"]
    assert isa_lint.lint_text(ok2, "t") == [] and isa_lint.lint_text(ok3, "t") == []
    assert [b[0] for b in isa_lint.lint_text(r3, "t")] == ["R3"]


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """bench.py: `--gpus N` and the number of ranks a launcher started must agree in EVERY combination (WORLD_SIZE=1 with --gpus 8
    used to run one GPU and print n_gpus = 1); with no launcher and N > 1 bench.py starts the N ranks itself -- here, without a
    GPU, both ranks refuse loudly and the parent hands their exit code on without printing a line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=300, cwd=ROOT)
    assert res.returncode != 0 and "must agree" in res.stderr and not res.stdout.strip()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="4", RANK="0"), timeout=300, cwd=ROOT)
    assert res.returncode != 0 and "must agree" in res.stderr
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.stderr.count("needs an MI355X") == 2, res.stderr[-2000:]          # one refusal per rank it started
