"""Headline benchmark: log-marginal-likelihood evaluations / second, N=50 000, d=3, RBF, fp64.

One "step" = one GP.log_likelihood(theta_t) with a NEW theta_t on data already resident in HBM:
covariance assembly (+noise) -> blocked Cholesky -> two triangular solves -> log-det -> scalar
(SURVEY 8d).  Usage (driver contract):

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts its N ranks itself, see launch_ranks)

N = 1: the fused single-GPU evaluation (fvgp_hip_loglik).  The line also carries, measured outside the timed
region, the other BASELINE.json configurations that fit one GPU ("configs") and the CPU restatement of the
reference path timed on this host ("cpu_baseline").

N > 1: ONE evaluation at a time, K+V row-sharded over the N ranks (block-cyclic 128-row blocks, RCCL all-gather of
the panel factors over xGMI: fvgp_amd/dist.py) -- the partitioning the north star names; "scaling": "strong".
The line also carries the collectives' bytes and GB/s per rank, the agreement with a single-GPU evaluation of the
same theta, and (side record, outside the timed region) the throughput of N independent replicas.
`--mode replicas` makes the replicas the headline instead ("scaling": "weak").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X dense fp64 matrix peak: 256 CU x 2.4 GHz x 128 flop/clk/CU
PEAK_HBM_GBS = 8000.0
XGMI_GBPS_PER_GPU = 7 * 153.0    # 7 links x ~153 GB/s (point-to-point)
TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01f_pmc_traffic.json")


def synth(n, d, seed=20240501):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(3.0 * np.sum(x, axis=1)) + 0.1 * rng.standard_normal(n)
    return x, y


# ------------------------------------------------------------------------------------------------
# CPU leg: the oracle (numpy/scipy restatement of the reference path) on this host
# ------------------------------------------------------------------------------------------------
def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def _host_ram_gb():
    try:
        import psutil
        return psutil.virtual_memory().total / 2 ** 30
    except Exception:
        return 0.0


def cpu_baseline(n_full, d, sample_n):
    """The oracle timed on this host.  If the host has the memory (>= 100 GB: K, K+V and the factor of N = n_full)
    and a short probe says the full size finishes in a few minutes, the FULL size is run once and reported as
    measured; otherwise a bounded sample is run and scaled with the stage exponents (assembly, addKV, solve: N^2;
    potrf: N^3), labelled as such."""
    from oracle import fvgp_oracle as orc
    threads = _blas_threads()
    theta = np.array([1.0] + [0.3] * d)

    last = {}

    def run(n):
        x, y = synth(n, d)
        t0 = time.perf_counter()
        ll, st = orc.log_likelihood_once(x, y, np.full(n, 0.01), theta, "rbf_ard")
        last.update(n=n, loglik=float(ll))          # kept: the HIP path is checked against it (headline_parity)
        return time.perf_counter() - t0, st

    def scaled(st, r):
        return (st["kmat"] + st["addKV"] + st["solve_logdet"]) * r ** 2 + st["potrf"] * r ** 3

    ram = _host_ram_gb()
    if sample_n <= 0:
        probe_n = 12000
        _, st = run(probe_n)
        est = scaled(st, n_full / probe_n)
        if ram >= 100.0 and est <= 300.0:
            wall, st = run(n_full)
            return {"value": 1.0 / wall, "unit": "evals/s", "cores": threads, "kind": "port",
                    "sample": (f"oracle (numpy/scipy restatement) on the FULL workload N={n_full}, d={d}, one evaluation: kmat "
                               f"{st['kmat']:.1f}s addKV {st['addKV']:.1f}s potrf {st['potrf']:.1f}s solve+logdet "
                               f"{st['solve_logdet']:.1f}s = {wall:.1f}s wall (BLAS threads {threads}, host cpus {os.cpu_count()}, "
                               f"RAM {ram:.0f} GB); measured, not extrapolated"),
                    "measured_seconds": wall, "extrapolated": False, "oracle_n": last["n"], "oracle_loglik": last["loglik"],
                    "oracle_theta": theta.tolist()}
        sample_n = 18000
    wall, st = run(sample_n)
    est = scaled(st, n_full / sample_n)
    return {"value": 1.0 / est, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": (f"oracle (numpy/scipy restatement) on N={sample_n}, d={d}: kmat {st['kmat']:.2f}s addKV "
                       f"{st['addKV']:.2f}s potrf {st['potrf']:.2f}s solve+logdet {st['solve_logdet']:.2f}s "
                       f"(wall {wall:.1f}s, BLAS threads {threads}, host cpus {os.cpu_count()}, RAM {ram:.0f} GB); scaled to "
                       f"N={n_full} with N^2 (assembly, addKV, solve) and N^3 (potrf) -> {est:.0f}s per evaluation"),
            "measured_seconds_at_sample": wall, "extrapolated": True, "oracle_n": last["n"], "oracle_loglik": last["loglik"],
            "oracle_theta": theta.tolist()}


def cpu_small_sizes(small):
    """CPU leg, the training loop's sizes: the oracle on this host at the size and theta of every `small_N` record (milliseconds of
    work; best of three), written beside the GPU number -- at C1's size (N=500) the line then shows GPU and CPU side by side."""
    from oracle import fvgp_oracle as orc
    for key, rec in small.items():
        if not isinstance(rec, dict) or "loglik_ms" not in rec:
            continue
        n, d = (int(t.split("=")[1]) for t in key.split())
        x, y = synth(n, d)
        ths = np.array([1.0] + [0.2 if d == 1 else 0.3] * d) * 1.01
        cpu, cpu_ll = 1e9, None
        for _ in range(3 if n <= 2000 else 1):
            t0 = time.perf_counter()
            cpu_ll, _ = orc.log_likelihood_once(x, y, np.full(n, 0.01), ths, "rbf_ard")
            cpu = min(cpu, time.perf_counter() - t0)
        rec.update(cpu_ms=1e3 * cpu, cpu_threads=_blas_threads(), cpu_over_gpu=1e3 * cpu / rec["loglik_ms"],
                   rel_diff_vs_oracle=abs(rec["loglik"] - cpu_ll) / abs(cpu_ll))


def headline_parity(cb, n, d, hip_at_n, device):
    """The HIP path against the log-likelihood the CPU leg's oracle run computed -- same synthetic inputs, same theta, outside the
    timed region.  When the oracle ran the full workload the HIP value is the one evaluated on the bench's own resident buffers
    (`hip_at_n`); when it ran a bounded sample the HIP path is evaluated once at that sample's size.  Tolerance: rel 1e-10
    (log-likelihood)."""
    from fvgp_amd import _lib
    theta, on = np.array(cb["oracle_theta"]), int(cb["oracle_n"])
    if on == n and hip_at_n is not None:
        hip = hip_at_n
    else:
        H = _lib.Handle(device)
        x, y = synth(on, d)
        npad = _lib.pad128(on)
        KV, alpha = H.empty(npad, npad), H.empty(npad, 1)
        hip, _, _, info = H.loglik(0, H.to_device(x), theta, H.to_device(np.full(on, 0.01)), H.to_device((y - np.mean(y)).reshape(on, 1)), KV, alpha)
        del KV, alpha
        H.close()
        if info != 0:
            hip = float("nan")
    ref = float(cb["oracle_loglik"])
    rel = abs(hip - ref) / abs(ref)
    return {"workload": f"N={on} d={d} RBF log_likelihood(theta), the bench's synthetic data" + ("" if on == n else " (the CPU leg's sample size)"),
            "theta": theta.tolist(), "hip": hip, "oracle": ref, "rel_diff": rel, "tolerance": 1e-10, "ok": bool(rel <= 1e-10)}


# ------------------------------------------------------------------------------------------------
# the other BASELINE.json configurations that fit one GPU (outside the timed region)
# ------------------------------------------------------------------------------------------------
class _stdout_to_stderr:
    """fd-level: what native libraries print while a communicator is built (RCCL's version banner goes to stdout) lands on stderr, so
    that the JSON line stays the only line on stdout"""
    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _progress(msg):
    """a line on stderr now and then: a run that is silent for minutes looks hung to a watchdog (the one JSON line stays alone on stdout)"""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def config_records():
    """C2 (N=20k RBF + posterior at P=1000), C3 (N=50k Matern-5/2 value + gradient), C5 (fvGP 4 x 10k, d=2) through
    the GP facade; wall time of the public call, best of two, against the fp64 MFMA bound of its flops."""
    import torch
    import warnings
    import fvgp_amd
    warnings.simplefilter("ignore")

    def best(f, reps=2):
        f()
        torch.cuda.synchronize()
        t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            t = min(t, time.perf_counter() - t0)
        return 1e3 * t

    peak = PEAK_FP64_MFMA_TFLOPS * 1e12
    out = {}
    th = np.array([1.0, 0.3, 0.3, 0.3])
    # C2
    n = 20000
    x, y = synth(n, 3)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    xp = np.random.default_rng(2).random((1000, 3))
    ll = best(lambda: gp.log_likelihood(th * 1.01))
    # the first posterior covariance after a new factorisation also inverts the factor's 2048 x 2048 diagonal blocks
    gp.posterior_covariance(xp)
    gp.set_hyperparameters(th * 1.005)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.posterior_covariance(xp)
    torch.cuda.synchronize()
    first_cov = 1e3 * (time.perf_counter() - t0)
    def cov_bound_ms(p):          # L^-1 k: n^2 p flop; S = kk - V^T V on the lower half: n p^2 flop
        return 1e3 * (float(n) * n * p + float(n) * p * p) / peak

    out["C2"] = {"workload": "N=20000 d=3 RBF: log_likelihood(theta); posterior mean / covariance at P=1000",
                 "loglik_ms": ll, "bound_ms": 1e3 * n ** 3 / 3 / peak, "frac": (1e3 * n ** 3 / 3 / peak) / ll,
                 "posterior_mean_ms": best(lambda: gp.posterior_mean(xp)),
                 "posterior_cov_ms": best(lambda: gp.posterior_covariance(xp)),
                 "posterior_cov_first_call_after_new_factor_ms": first_cov,
                 "posterior_cov_bound_ms": 1e3 * (n * n * 1000.0 + 2.0 * n * 1000.0 ** 2) / peak}
    out["C2"]["posterior_cov_frac"] = out["C2"]["posterior_cov_bound_ms"] / out["C2"]["posterior_cov_ms"]
    # a state refresh (set_hyperparameters: evaluation into the state's buffers + the inverted diagonal blocks the posterior's sweep
    # substitutes with, enqueued behind it since round 6), synchronised: what "first call after a new factor" no longer pays itself
    out["C2"]["state_refresh_ms"] = best(lambda: gp.set_hyperparameters(th * 1.005))
    # variance_only in the default 'Chol' mode still forms S, as the reference does (gp_posterior.py:246); many points go through the
    # device in chunks of 1024 with S assembled on the host block row by block row (fvgp_amd/gp.py _posterior_chunked)
    out["C2"]["posterior_cov_variance_only_ms"] = best(lambda: gp.posterior_covariance(xp, variance_only=True))
    xp4 = np.random.default_rng(3).random((4000, 3))
    out["C2"]["posterior_cov_p4000_ms"] = best(lambda: gp.posterior_covariance(xp4))
    out["C2"]["posterior_cov_p4000_bound_ms"] = cov_bound_ms(4000)
    out["C2"]["posterior_cov_p4000_frac"] = cov_bound_ms(4000) / out["C2"]["posterior_cov_p4000_ms"]
    out["C2"]["posterior_cov_p4000_chunk_groups"] = gp._posterior_groups
    del gp
    torch.cuda.empty_cache()
    _progress("C2 done")
    # the training loop's sizes (gp_mcmc.py:96-224 calls log_likelihood(theta) 10 000 times; C1 is N=500, d=1): wall time of the
    # public call, best of five
    small = {}
    for n, d in ((500, 1), (2000, 3), (4000, 3), (8000, 3), (12000, 3)):
        x, y = synth(n, d)
        ths = np.array([1.0] + [0.2 if d == 1 else 0.3] * d)
        gp = fvgp_amd.GP(x, y, init_hyperparameters=ths, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
        ms = best(lambda: gp.log_likelihood(ths * 1.01), reps=5)
        small[f"N={n} d={d}"] = {"loglik_ms": ms, "bound_ms": 1e3 * n ** 3 / 3 / peak, "frac": (1e3 * n ** 3 / 3 / peak) / ms}
        small[f"N={n} d={d}"]["loglik"] = gp.log_likelihood(ths * 1.01)       # (checked against the oracle by the CPU leg)
        del gp
    out["small_N"] = {"workload": "RBF log_likelihood(theta) at the training loop's sizes (C1: N=500 d=1)", **small}
    torch.cuda.empty_cache()
    _progress("small sizes done")
    # C3
    n = 50000
    x, y = synth(n, 3)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=th, noise_variances=np.full(n, 0.01), kernel_function="matern52_ard")
    vg = best(lambda: gp.neg_log_likelihood_gradient(th * 1.01), reps=1)
    out["C3"] = {"workload": "N=50000 d=3 Matern-5/2: log-marginal-likelihood + hyperparameter gradient (POTRF + POTRI + fused trace)",
                 "value_grad_ms": vg, "bound_ms": 1e3 * float(n) ** 3 / peak, "frac": (1e3 * float(n) ** 3 / peak) / vg,
                 "loglik_ms": best(lambda: gp.log_likelihood(th * 1.01), reps=1)}
    del gp
    torch.cuda.empty_cache()
    _progress("C3 done")
    # C5
    rng = np.random.default_rng(20240501)
    xm = rng.random((10000, 2))
    s = xm.sum(axis=1)
    ym = np.stack([np.sin(3 * s), np.cos(3 * s), np.linalg.norm(xm, axis=1), np.sin(3 * s) * np.cos(3 * s)], axis=1)
    ym = ym + 0.1 * rng.standard_normal(ym.shape)
    th5 = np.array([1.0, 0.3, 0.3, 1.0])
    gp = fvgp_amd.fvGP(xm, ym, init_hyperparameters=th5, noise_variances=np.full(ym.shape, 0.01))
    n = 40000
    ll = best(lambda: gp.log_likelihood(th5 * 1.01))
    out["C5"] = {"workload": "fvGP 4 tasks x N=10000 d=2 (index set 40000 x 3), default Matern-3/2 ARD kernel: log_likelihood(theta)",
                 "loglik_ms": ll, "bound_ms": 1e3 * float(n) ** 3 / 3 / peak, "frac": (1e3 * float(n) ** 3 / 3 / peak) / ll}
    del gp
    torch.cuda.empty_cache()
    _progress("C5 done")
    # C4's size on ONE GPU (the 8-GPU run is the driver's): the fused single-GPU evaluation against the row-sharded driver at one
    # rank, its whole multi-rank code path taken with the collectives issued through RCCL (one-rank communicator)
    try:
        from fvgp_amd import _lib
        from fvgp_amd.device import default_handle
        from fvgp_amd.dist import ShardedGP
        n = 100000
        x, y = synth(n, 3)
        nv = np.full(n, 0.01)
        H = default_handle()
        npad = _lib.pad128(n)
        KV = H.empty(npad, npad)
        xd, vd, ymd = H.to_device(x), H.to_device(nv), H.to_device((y - np.mean(y)).reshape(n, 1))
        vals = []
        fused = best(lambda: vals.append(H.loglik(0, xd, th * 1.01, vd, ymd, KV, None)[0]), reps=1)
        del KV
        torch.cuda.empty_cache()
        with _stdout_to_stderr():
            sh = ShardedGP(x, y, nv, kernel="rbf_ard", panel=2048, rank=0, world=1, force_collectives=True, collectives="rccl")
        svals = []
        sharded = best(lambda: svals.append(sh.log_likelihood(th * 1.01)[0]), reps=1)
        # the collectives of ONE more evaluation, timed with events on the chain stream, i.e. beside the rank's trailing update
        # (fvgp_hip_comm_profile); the same calls alone on the idle chip: profiles/r04_rccl_beside_update.txt
        sh.ops.set_option("profile", 1)
        sh.collective_summary()
        sh.log_likelihood(th * 1.01)
        torch.cuda.synchronize()
        cs = sh.collective_summary()
        sh.ops.set_option("profile", 0)
        flops = float(n) ** 3 / 3
        out["C4_size_one_gpu"] = {
            "workload": "N=100000 d=3 RBF log_likelihood(theta) on ONE MI355X: fused driver vs the row-sharded driver at one rank "
                        "(2048-wide panels; panel buffers, diagonal-block gather and panel all-gather through a one-rank RCCL communicator)",
            "fused_ms": fused, "fused_tflops": flops / fused / 1e9, "sharded_world1_rccl_ms": sharded,
            "sharded_world1_rccl_tflops": flops / sharded / 1e9, "bound_ms": 1e3 * flops / peak,
            "rel_diff": abs(vals[-1] - svals[-1]) / abs(vals[-1]),
            "collectives": {k: {"calls": v[0], "bytes_from_peers": v[1], "ms_on_chain_stream_beside_update": v[2]} for k, v in cs.items()}}
        del sh
        torch.cuda.empty_cache()
        _progress("C4 size on one GPU done")
        # what ONE rank of the configured 8-rank run has to do, on this GPU: rank 0's kernel / stream schedule of an 8-rank
        # evaluation at N = 100k with the collectives replaced by local copies of the same size (tools/shard_emulate.py): the
        # compute-side floor of C4 before any byte crosses xGMI, against the rank's share of the flops at the fp64 MFMA peak
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from shard_emulate import emulate
            ms8, _, enq8 = emulate(n, world=8, rank=0, panel=2048, steps=2)
            out["C4_size_one_gpu"]["p8_rank0_schedule_emulated_ms"] = ms8
            out["C4_size_one_gpu"]["p8_rank_flop_bound_ms"] = 1e3 * flops / 8 / peak
            out["C4_size_one_gpu"]["p8_frac_of_flop_bound"] = (1e3 * flops / 8 / peak) / ms8
            out["C4_size_one_gpu"]["p8_host_enqueue_ms"] = enq8
        except Exception as e:                  # noqa: BLE001
            out["C4_size_one_gpu"]["p8_emulation_error"] = f"{type(e).__name__}: {e}"[:200]
    except Exception as e:                      # noqa: BLE001 -- a side record must not take the headline down
        out["C4_size_one_gpu"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


# ------------------------------------------------------------------------------------------------
def single_gpu_eval(H, _lib, x, y, thetas):
    """fused single-GPU evaluations of the given thetas on this rank (replica leg / cross-check); returns (values, seconds)"""
    import torch
    n = len(x)
    npad = _lib.pad128(n)
    xd, vd = H.to_device(x), H.to_device(np.full(n, 0.01))
    ymd = H.to_device((y - np.mean(y)).reshape(n, 1))
    KV, alpha = H.empty(npad, npad), H.empty(npad, 1)
    vals = []
    H.loglik(0, xd, thetas[0], vd, ymd, KV, alpha)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for th in thetas:
        ll, _, _, info = H.loglik(0, xd, th, vd, ymd, KV, alpha)
        if info != 0:
            raise SystemExit(f"single-GPU evaluation failed: info={info}")
        vals.append(ll)
    torch.cuda.synchronize()
    return vals, time.perf_counter() - t0


def sharded_main(args, x, y, world, rank, local, dist):
    """N > 1 headline: one evaluation at a time row-sharded over the ranks."""
    import torch
    from fvgp_amd import _lib
    from fvgp_amd.dist import ShardedGP
    n, d = args.n, args.d
    theta0 = np.array([1.0] + [0.3] * d)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    done = threading.Event()
    state = {"replicas": None}         # filled before the first data-path collective: what a hung run still has to report

    def watchdog():
        if not done.wait(args.sharded_timeout):
            err = f"the row-sharded evaluation's collectives did not complete within {args.sharded_timeout:.0f} s"
            rep = state["replicas"]
            if state.get("completed"):
                # one collective backend has been through its whole timed region (every rank knows: the decision was a collective);
                # the one tried after it hangs: the completed line stands, with what happened to the other
                if rank == 0:
                    line = state["completed"]
                    line["collectives_ab"][state.get("in_flight", "?")] = {"error": err}
                    print(json.dumps(line), flush=True)
                os._exit(0)
            if rank == 0:
                # the headline of an N > 1 run is ONE evaluation sharded over the ranks: when that did not complete the line says
                # value 0.0 + error, the replicas measured before it ride along as a side record, and the exit code is non-zero
                line = {"metric": "log_marginal_likelihood_evals_per_sec", "value": 0.0, "unit": "evals/s",
                        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": f"N={n} d={d} RBF(ARD) log_likelihood(theta): K-assembly+noise, Cholesky, 2 triangular "
                                               f"solves, log-det", "n": n, "d": d, "kernel": "rbf_ard",
                                   "parallelism": f"one evaluation row-sharded over {world} GPUs (block-cyclic 128-row blocks)"},
                        "error": err}
                if rep is not None:
                    line["replicas"] = {"value": rep["value"], "unit": "evals/s", "scaling": "weak", "evals_per_gpu": rep["evals_per_gpu"],
                                        "ms_per_step": rep["ms_per_step"],
                                        "parallelism": f"{world} independent replicas (one theta stream per GPU), measured before the sharded attempt"}
                print(json.dumps(line), flush=True)
            os._exit(3)                # a hung sharded evaluation is a failed run, whatever else was measured

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return _sharded_body(args, x, y, world, rank, local, dist, sync_all, theta0, n, d, state)
    finally:
        done.set()


def _sharded_body(args, x, y, world, rank, local, dist, sync_all, theta0, n, d, state):
    import torch
    # panel width of the row-sharded driver: 2048 from N = 40 000 on -- the per-rank schedule is as fast as with 1024 (emulated:
    # profiles/r06_shard_emulation.txt: P=8 N=50k 94.2 vs 94.1 ms, N=100k 618 vs 626) and an evaluation issues half as many
    # collectives, each of which starts 0.34 ms late beside the update
    panel = args.outer_block or (2048 if n >= 40000 else 1024)
    from fvgp_amd import _lib
    from fvgp_amd.dist import ShardedGP
    # side record + cross-check, measured FIRST (no data-path collective: if the sharded run below hangs on this node, the
    # watchdog still has a line to print): every rank evaluates its own theta stream on its own GPU (replicas); the first
    # theta is the last theta of the sharded run
    theta_last = theta0 * (1.0 + 0.02 * (args.warmup + args.steps - 1))
    Hs = _lib.Handle(local)
    thetas = [theta_last] + [theta0 * (1.0 + 0.02 * (100 + rank + world * t)) for t in range(max(1, min(2, args.steps)))]
    vals, secs = single_gpu_eval(Hs, _lib, x, y, thetas)
    Hs.close()
    del Hs
    torch.cuda.empty_cache()
    if dist is not None:
        tt = torch.tensor([secs], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        secs = float(tt.item())
    state["replicas"] = {"value": world * len(thetas) / secs, "evals_per_gpu": len(thetas), "ms_per_step": 1e3 * secs / len(thetas)}
    # world == 1 (`--gpus 1 --mode sharded`): the multi-rank code path on one GPU, its collectives issued through RCCL
    # (a one-rank communicator) unless --backend / --collectives say otherwise
    force = dist is None
    base = "rccl" if args.backend == "nccl" else "torch"
    candidates = [base, "ipc"] if args.collectives == "auto" else [args.collectives]
    if world == 1 and args.collectives == "auto":
        candidates = [base]                                        # one rank has no peer to pull from: nothing to compare

    def all_ranks_ok(ok):
        """True when `ok` holds on EVERY rank (a collective of its own: every rank decides the same way)"""
        if dist is None:
            return bool(ok)
        tt = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) == 0.0

    def max_over_ranks(v):
        if dist is None:
            return float(v)
        tt = torch.tensor([v], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def build(kind):
        with _stdout_to_stderr():
            return ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel=panel,
                             rank=None if dist is not None else 0, world=None if dist is not None else 1,
                             force_collectives=force, collectives=kind)

    def coll_record(gp, evals):
        """the collectives of the last `evals` evaluations (events on the chain stream around every call): per kind the calls, the
        bytes this rank received, the milliseconds, and the rate per peer link against the 153 GB/s of one xGMI link -- a direct
        gather pulls from all world - 1 peers at once, a ring moves everything over one link: both are priced per link"""
        rec = {}
        peers = max(world - 1, 1)
        for kind, (calls, nbytes, ms) in gp.collective_summary().items():
            if calls:
                gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 and nbytes > 0 else None
                rec[kind] = {"calls_per_eval": calls / evals, "bytes_received_per_rank_per_eval": nbytes / evals,
                             "ms_on_chain_stream_per_eval": ms / evals, "ms_per_call": ms / calls,
                             "GBps_per_rank": gbps, "GBps_per_peer_link": gbps / peers if gbps else None,
                             "frac_of_one_xgmi_link_153": gbps / peers / 153.0 if gbps else None,
                             "frac_of_xgmi_7x153": gbps / XGMI_GBPS_PER_GPU if gbps else None}
        return rec

    def make_line(collectives_via, meas, trial):
        elapsed, prof, coll, ll, comm = meas["elapsed"], meas["prof"], meas["coll"], meas["ll"], meas["comm"]
        syrk_tflops = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
        whole = args.steps * (n ** 3) / 3.0 / elapsed / 1e12
        out = {
            "metric": "log_marginal_likelihood_evals_per_sec", "value": args.steps / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={n} d={d} RBF(ARD) log_likelihood(theta): K-assembly+noise, Cholesky, forward solve, log-det; "
                                   f"ONE evaluation row-sharded over the GPUs", "n": n, "d": d, "kernel": "rbf_ard",
                       "parallelism": f"block-cyclic 128-row blocks over {world} GPUs; per {panel}-wide panel: all-gather of the "
                                      f"diagonal block from its owners, all-gather of the panel factor over xGMI (by: collectives_via), one panel of look-ahead"},
            "cholesky_tflops": whole, "cholesky_tflops_per_gpu": whole / world,
            "cholesky_frac_of_fp64_mfma_peak": whole / world / PEAK_FP64_MFMA_TFLOPS,
            "loglik_last": ll,
            "rel_diff_vs_single_gpu": abs(ll - vals[0]) / abs(vals[0]),
            "single_gpu_loglik_at_same_theta": vals[0],
            "collectives": coll, "collectives_via": {"rccl": "RCCL called from libfvgp_hip.so on the chain stream (fvgp_hip_comm_init)",
                                                     "ipc": "direct pulls between IPC-mapped windows on the copy engines (csrc/ipc.hip, fvgp_hip_comm_init_ipc)",
                                                     "torch": "torch.distributed callbacks (gloo test path)"}[collectives_via],
            "collectives_chosen": collectives_via,
            # what the communicator says about itself: ncclCommCount / ncclCommUserRank / ncclCommCuDevice for RCCL (not what this
            # script asked for), the bound rank count for the direct collectives
            "communicator": comm,
            "ranks_seen_by_communicator": comm.get("nccl_comm_count", comm.get("nranks_bound")),
            # --collectives auto: the whole timed region through each backend (ms per evaluation, per-collective timings, or the error)
            "collectives_ab": trial,
            "xgmi_peak_GBps_per_gpu": XGMI_GBPS_PER_GPU, "xgmi_GBps_per_link": 153.0,
            "roofline": {"kernel": "gemm_f64_kernel<0, 0, 1> (row-sharded trailing update, rank 0's launches)", "bound": "mfma",
                         "achieved": syrk_tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": syrk_tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                         "launches": prof["launches"], "avg_launch_ms": prof["ms"] / max(prof["launches"], 1.0),
                         "algorithmic_flops_per_launch": prof["flops"] / max(prof["launches"], 1.0)},
            "replicas": {"value": world * len(thetas) / secs, "unit": "evals/s", "scaling": "weak",
                         "note": f"{world} independent single-GPU evaluations at a time (one theta stream per GPU), "
                                 f"{len(thetas)} per GPU, outside the timed region"},
        }
        return out
    # --collectives auto: warm-up + the WHOLE timed region through each candidate, the process group's own collectives first (the
    # known path: its line exists before the direct pulls are tried, and the watchdog prints it if they hang), timed the way the
    # contract says (barrier + synchronise on both sides, max over ranks); the faster one is the line's headline, both are recorded
    trial, lines = {}, {}
    for kind in candidates:
        rec = {}
        state["in_flight"] = kind
        if kind == "ipc" and args.inject_ipc_hang:
            time.sleep(1e6)                                         # (test hook: the watchdog has the completed backend's line)
        try:
            g = build(kind)
            ok = True
        except Exception as e:                                      # noqa: BLE001 -- recorded; the other candidate still runs
            g, ok = None, False
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        if not all_ranks_ok(ok):
            rec.setdefault("error", "communicator construction failed on another rank")
            trial[kind] = rec
            if g is not None:
                g.close()
            continue
        meas = None
        try:
            comm = g.comm_info()
            if dist is not None:
                comm["torch_distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
            g.log_likelihood(theta0)                                # first call: scratch growth, tile tables
            for t in range(args.warmup):
                g.log_likelihood(theta0 * (1.0 + 0.02 * t))
            H = g.ops
            H.set_option("profile", 1)
            H.get_profile()
            g.collective_summary()
            sync_all()
            t0 = time.perf_counter()
            for t in range(args.steps):
                ll, logdet, quad = g.log_likelihood(theta0 * (1.0 + 0.02 * (args.warmup + t)))
            sync_all()
            elapsed = max_over_ranks(time.perf_counter() - t0)
            meas = {"elapsed": elapsed, "prof": H.get_profile(), "coll": coll_record(g, args.steps), "ll": ll, "comm": comm}
            H.set_option("profile", 0)
            rec.update(ms_per_eval=1e3 * elapsed / args.steps, collectives=meas["coll"], loglik=ll)
            ok = True
        except Exception as e:                                      # noqa: BLE001
            ok = False
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        if not all_ranks_ok(ok):
            rec.setdefault("error", "the evaluation failed on another rank")
            rec.pop("ms_per_eval", None)
            meas = None
        trial[kind] = rec
        g.close()                                                   # one communicator at a time on the handle
        del g
        torch.cuda.empty_cache()
        if meas is not None:
            lines[kind] = meas
            best = min(lines, key=lambda q: lines[q]["elapsed"])
            state["completed"] = make_line(best, lines[best], dict(trial))
    if not lines:
        raise RuntimeError("no collective backend completed an evaluation: " + json.dumps(trial))
    chosen = min(lines, key=lambda q: lines[q]["elapsed"])
    if rank == 0:
        print(json.dumps(make_line(chosen, lines[chosen], trial)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): THIS process -- which has not imported torch and has
    not touched a GPU -- starts the N ranks as children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun would set them,
    rendezvous on 127.0.0.1), relays rank 0's single JSON line on stdout (everything else the ranks print goes to stderr) and
    returns the worst exit code.  A rank that dies takes the others with it after a grace period: no orphan keeps a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", FVGP_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    first_failure = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed and first_failure is None:
            first_failure = time.time()
        if first_failure is not None and time.time() - first_failure > 60.0:
            for p in procs:                       # exactly the processes started here
                if p.poll() is None:
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
    t.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    worst = next((c for c in codes if c not in (0, None)), 0)
    if worst:
        print(f"[bench] rank exit codes {codes}", file=sys.stderr, flush=True)
    return worst if worst > 0 else (128 - worst if worst < 0 else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--npoints", dest="n", type=int, default=50000, help="data points (use --npoints under torchrun: its parser claims --n)")
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--outer-block", type=int, default=0, help="K of the trailing SYRK (0 = library default)")
    ap.add_argument("--lookahead", type=int, default=-1, help="1/0: factor the next panel on a side stream (-1 = library default)")
    ap.add_argument("--mode", choices=["auto", "replicas", "sharded"], default="auto",
                    help="N>1: ONE evaluation row-sharded over the GPUs (auto / sharded: block-cyclic rows, RCCL all-gather of panel "
                         "factors; strong scaling) or independent replicas (one theta stream per GPU; weak scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for tests that "
                                                      "put several ranks on one GPU)")
    ap.add_argument("--collectives", choices=["auto", "rccl", "ipc", "torch"], default="auto",
                    help="N>1: who moves the panel factors.  rccl = RCCL called from the library on the chain stream; ipc = direct pulls "
                         "between IPC-mapped windows on the copy engines (csrc/ipc.hip); torch = torch.distributed callbacks (gloo "
                         "tests); auto = the whole timed region through the backend's own (rccl for nccl, torch for gloo) AND through "
                         "ipc, both recorded in the line, the faster one its headline")
    ap.add_argument("--sharded-timeout", type=float, default=600.0)
    ap.add_argument("--cpu-sample-n", type=int, default=0, help="0 = the full workload if the host can hold it, else N=18000 scaled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inject-sharded-failure", action="store_true", help=argparse.SUPPRESS)   # tests: the sharded run raises on every rank
    ap.add_argument("--inject-ipc-hang", action="store_true", help=argparse.SUPPRESS)          # tests: the direct-pull candidate never returns
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C3 / C5 records measured after the timed region")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        ap.error("--steps must be >= 1 and --warmup >= 0")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        # no launcher around this process.  --gpus 1: run here.  --gpus N > 1: start the N ranks as children of this process,
        # which has neither imported torch nor touched a GPU (never an exec from a process that has)
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
        world, rank = 1, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        if args.gpus != world:
            # a launcher started `world` ranks: the line would say n_gpus = world whatever --gpus asked for
            raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: the two must agree")
    import torch
    from fvgp_amd import _lib

    local = int(os.environ.get("FVGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # FVGP_DEVICE: ranks sharing a GPU (tests)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: device {local} asked for, {torch.cuda.device_count()} visible (--gpus {args.gpus} needs that many GPUs on this node)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            with _stdout_to_stderr():
                dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend=args.backend)

    n, d = args.n, args.d
    x, y = synth(n, d)
    sharded_error = None
    if args.mode == "sharded" or (args.mode == "auto" and world > 1):
        try:
            if args.inject_sharded_failure:
                raise RuntimeError("injected failure of the row-sharded evaluation (test hook)")
            return sharded_main(args, x, y, world, rank, local, dist)
        except Exception as e:              # noqa: BLE001 -- an exception (not a hang: the watchdog owns those) on the sharded path
            if args.mode == "sharded" or dist is None:
                raise
            # the same code runs on every rank, so every rank lands here: the run still reports the replicas record
            # (independent evaluations, no data-path collective) and says what happened to the sharded one
            import traceback
            traceback.print_exc()
            sharded_error = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    H = _lib.Handle(local)
    if args.outer_block:
        H.set_option("outer_block", args.outer_block)
    if args.lookahead >= 0:
        H.set_option("lookahead", args.lookahead)
    npad = _lib.pad128(n)
    xd = H.to_device(x)
    vd = H.to_device(np.full(n, 0.01))
    ymd = H.to_device((y - np.mean(y)).reshape(n, 1))
    KV = H.empty(npad, npad)
    alpha = H.empty(npad, 1)
    theta0 = np.array([1.0] + [0.3] * d)

    def theta_at(t):
        # every rank walks its own slice of the proposal sequence
        return theta0 * (1.0 + 0.02 * (t * world + rank))

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.warmup):
        H.loglik(0, xd, theta_at(t), vd, ymd, KV, alpha)
    H.set_option("profile", 1)
    prof = {"launches": 0.0, "ms": 0.0, "flops": 0.0, "potrf_ms": 0.0, "kmat_ms": 0.0, "kmat_bytes": 0.0, "tail_ms": 0.0, "bytes": 0.0}
    sync_all()
    t0 = time.perf_counter()
    for t in range(args.steps):
        ll, logdet, quad, info = H.loglik(0, xd, theta_at(args.warmup + t), vd, ymd, KV, alpha)
        if info != 0 or not np.isfinite(ll):
            raise SystemExit(f"evaluation {t} failed: info={info} loglik={ll}")
        p = H.get_profile()
        for k in prof:
            prof[k] += p[k]
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    H.set_option("profile", 0)
    _progress(f"timed region done: {1e3 * elapsed / args.steps:.1f} ms per step")
    # outside the timed region: one evaluation at the theta the CPU leg's oracle runs, on the same resident buffers (headline_parity)
    hip_theta0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        hip_theta0, _, _, info0 = H.loglik(0, xd, theta0, vd, ymd, KV, alpha)
        if info0 != 0:
            hip_theta0 = float("nan")

    out = None
    if rank == 0:
        traffic, traffic_source = None, None
        for name in TRAFFIC_FILES:
            tfile = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tfile) and n == 50000:
                # HBM-side bytes per trailing-update launch from the committed rocprofv3 --pmc passes of this same
                # command (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE); not re-measured in this run
                traffic = json.load(open(tfile)).get("bytes_per_launch")
                traffic_source = f"committed rocprofv3 --pmc passes of this command (profiles/{name}); not measured live"
                break
        evals = args.steps * world
        ms_per_step = 1e3 * elapsed / args.steps
        syrk_tflops = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
        potrf_tflops = (args.steps * (n ** 3) / 3.0) / (prof["potrf_ms"] * 1e-3) / 1e12 if prof["potrf_ms"] > 0 else 0.0
        out = {
            "metric": "log_marginal_likelihood_evals_per_sec", "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            # one evaluation at a time: at N > 1 the SAME evaluation is row-sharded over the ranks (total work fixed: "strong");
            # --mode replicas gives every GPU its own theta stream instead (work grows with N: "weak")
            "higher_is_better": True, "scaling": "weak" if (world > 1 or args.mode == "replicas") else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={n} d={d} RBF(ARD) log_likelihood(theta): K-assembly+noise, Cholesky, "
                                   f"2 triangular solves, log-det", "n": n, "d": d, "kernel": "rbf_ard",
                       "parallelism": "1 GPU" if world == 1 else f"{world} independent replicas (one theta stream per GPU)"},
            "cholesky_tflops": potrf_tflops,
            "cholesky_frac_of_fp64_mfma_peak": potrf_tflops / PEAK_FP64_MFMA_TFLOPS,
            "loglik_last": ll,
            # the other stages of the evaluation, HIP events on the launch stream inside the timed region
            "k_assembly": {"kernel": "kmat_kernel (pairwise distance + RBF, + noise on the diagonal; lower 128-tiles written once)",
                           "bound": "hbm", "achieved": prof["kmat_bytes"] / (prof["kmat_ms"] * 1e-3) / 1e9 if prof["kmat_ms"] > 0 else 0.0,
                           "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": prof["kmat_bytes"] / (prof["kmat_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS if prof["kmat_ms"] > 0 else 0.0,
                           "ms_per_eval": prof["kmat_ms"] / args.steps, "algorithmic_bytes_per_eval": prof["kmat_bytes"] / args.steps},
            "after_factorisation_ms_per_eval": prof["tail_ms"] / args.steps,
            "roofline": {
                "kernel": "gemm_f64_kernel<0, 0, 1> (trailing update of the blocked Cholesky, lower tiles)",
                "bound": "mfma", "achieved": syrk_tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": syrk_tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                "launches": prof["launches"], "avg_launch_ms": prof["ms"] / max(prof["launches"], 1.0),
                "algorithmic_flops_per_launch": prof["flops"] / max(prof["launches"], 1.0),
                # every C tile of the launch read and written once + the panel's rows read once (csrc/api.hip lower_bytes); the
                # kernel is MFMA-bound: `traffic` above this is operand re-fetch past the L2, at ~1.4 TB/s of an 8 TB/s fabric
                "algorithmic_bytes_per_launch": prof["bytes"] / max(prof["launches"], 1.0),
                "traffic_over_algorithmic": (traffic / (prof["bytes"] / max(prof["launches"], 1.0))) if traffic and prof["bytes"] > 0 else None,
            },
        }
        out["k_assembly"]["frac_of_achievable_6290"] = out["k_assembly"]["achieved"] / 6290.0   # the guide's measured streaming rate
    del KV, alpha
    H.close()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        if not args.no_configs and n == 50000:
            try:
                out["configs"] = config_records()
            except Exception as e:                                  # noqa: BLE001 -- reported in the line, the headline stands
                out["configs"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline:                                # the CPU leg is timed on rank 0 at N=1 only
            _progress("configs done; the CPU leg (the oracle on this host, about a minute and a half at N = 50000) starts")
            out["cpu_baseline"] = cpu_baseline(n, d, args.cpu_sample_n)
            if isinstance(out.get("configs", {}).get("small_N"), dict):
                cpu_small_sizes(out["configs"]["small_N"])
            _progress("CPU leg done")
            out["headline_parity"] = headline_parity(out["cpu_baseline"], n, d, hip_theta0, local)
    if rank == 0:
        if sharded_error is not None:
            # same shape as the watchdog's line: the headline of an N > 1 run is ONE evaluation sharded over the ranks; it raised, so
            # the line says value 0.0 + error and the replicas measured here ride along as a side record (exit code 4 below)
            out = {"metric": out["metric"], "value": 0.0, "unit": out["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": out["config"]["workload"], "n": n, "d": d, "kernel": "rbf_ard",
                              "parallelism": f"one evaluation row-sharded over {world} GPUs (block-cyclic 128-row blocks)"},
                   "error": ("the row-sharded evaluation raised: " + sharded_error)[:600],
                   "replicas": {"value": out["value"], "unit": out["unit"], "scaling": "weak", "ms_per_step": out["ms_per_step"],
                                "parallelism": out["config"]["parallelism"], "roofline": out["roofline"]}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None and not out.get("headline_parity", {"ok": True})["ok"]:
        raise SystemExit(5)                 # the metric's own workload disagrees with the oracle: the number above does not count
    if sharded_error is not None:
        # the sharded headline the launch asked for did not complete: the launcher must not read a clean exit
        raise SystemExit(4)


if __name__ == "__main__":
    main()
