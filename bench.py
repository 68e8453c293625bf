"""Headline benchmark: log-marginal-likelihood evaluations / second, N=50 000, d=3, RBF, fp64.

One "step" = one GP.log_likelihood(theta_t) with a NEW theta_t on data already resident in HBM:
covariance assembly (+noise) -> blocked Cholesky -> two triangular solves -> log-det -> scalar
(SURVEY 8d).  Usage (driver contract):

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: every rank evaluates its own theta sequence on its own GPU (independent replicas of the
population of hyperparameter proposals a trainer evaluates -- SURVEY 8e "replicas-only"); no
data-path collective, "scaling": "weak".  After the timed region the same workload is also run as ONE
evaluation row-sharded over the ranks (block-cyclic rows, RCCL all-gather of panel factors) and reported
under "sharded"; `--mode sharded` makes that the headline instead.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X dense fp64 matrix peak: 256 CU x 2.4 GHz x 128 flop/clk/CU
PEAK_HBM_GBS = 8000.0


def synth(n, d, seed=20240501):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(3.0 * np.sum(x, axis=1)) + 0.1 * rng.standard_normal(n)
    return x, y


def cpu_baseline(n_full, d, sample_n):
    """The oracle (numpy/scipy restatement of the reference path) timed on this host on a bounded
    sample, scaled to N = n_full with the stage exponents (assembly, addKV, solve: N^2; potrf: N^3)."""
    from oracle import fvgp_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    x, y = synth(sample_n, d)
    nv = np.full(sample_n, 0.01)
    theta = np.array([1.0] + [0.3] * d)
    t0 = time.perf_counter()
    _, st = orc.log_likelihood_once(x, y, nv, theta, "rbf_ard")
    wall = time.perf_counter() - t0
    r = n_full / sample_n
    est = (st["kmat"] + st["addKV"] + st["solve_logdet"]) * r ** 2 + st["potrf"] * r ** 3
    return {
        "value": 1.0 / est, "unit": "evals/s", "cores": int(threads), "kind": "port",
        "sample": (f"oracle (numpy/scipy restatement) on N={sample_n}, d={d}: kmat {st['kmat']:.2f}s addKV "
                   f"{st['addKV']:.2f}s potrf {st['potrf']:.2f}s solve+logdet {st['solve_logdet']:.2f}s "
                   f"(wall {wall:.1f}s, BLAS threads {threads}, host cpus {os.cpu_count()}); scaled to N={n_full} "
                   f"with N^2 (assembly, addKV, solve) and N^3 (potrf) -> {est:.0f}s per evaluation"),
        "measured_seconds_at_sample": wall,
    }


def sharded_measure(args, x, y, world, rank, local, dist, steps, warmup, check_theta=None, check_value=None):
    """One evaluation at a time, K+V row-sharded over the ranks (fvgp_amd/dist.py).  Returns (on every rank)
    the timing of `steps` evaluations after `warmup`, max over ranks."""
    import torch
    from fvgp_amd.dist import ShardedGP
    n, d = args.n, args.d
    gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel=args.outer_block or 1024,
                   rank=rank if dist is not None else 0, world=world if dist is not None else 1)
    theta0 = np.array([1.0] + [0.3] * d)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(warmup):
        gp.log_likelihood(theta0 * (1.0 + 0.02 * t))
    sync_all()
    t0 = time.perf_counter()
    for t in range(steps):
        ll, logdet, quad = gp.log_likelihood(theta0 * (1.0 + 0.02 * (warmup + t)))
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gathered = 0.0                                                   # bytes every rank receives per evaluation
    if world > 1:
        for J in range(gp.npan - 1):
            k = (gp.nb_max - gp.bnd[J + 1] // 128 // world) * 128
            gathered += 8.0 * (world - 1) * k * (gp.bnd[J + 1] - gp.bnd[J])
    extra = {}
    if world > 1:                                                    # one more (untimed) evaluation with events around the collectives
        gp.collective_events = []
        gp.log_likelihood(theta0)
        for kind, (calls, nbytes, ms) in gp.collective_summary().items():
            extra[kind] = {"calls": calls, "bytes_received_per_rank": nbytes, "ms_on_chain_stream": ms,
                           "GBps_per_rank": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None}
        gp.collective_events = None
    if check_theta is not None:                                      # same theta as a single-GPU evaluation of this run
        ll_c, _, _ = gp.log_likelihood(check_theta)
        if check_value is not None:
            extra.update({"loglik_at_check_theta": ll_c, "single_gpu_loglik_at_check_theta": check_value,
                          "rel_diff_vs_single_gpu": abs(ll_c - check_value) / abs(check_value)})
    return {**extra, "evals_per_s": steps / elapsed, "ms_per_eval": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
            "tflops_per_gpu": steps * (n ** 3) / 3.0 / elapsed / 1e12 / world, "loglik_last": ll,
            "panel": gp.NB, "all_gather_bytes_per_rank_per_eval": gathered,
            "parallelism": f"block-cyclic 128-row blocks over {world} GPU(s), per panel: all-reduce of the diagonal "
                           f"block, RCCL all-gather of the panel factor, one panel of look-ahead"}


def sharded_main(args, x, y, world, rank, local, dist):
    n, d = args.n, args.d
    r = sharded_measure(args, x, y, world, rank, local, dist, args.steps, args.warmup)
    if rank == 0:
        out = {
            "metric": "log_marginal_likelihood_evals_per_sec", "value": r["evals_per_s"], "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_eval"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={n} d={d} RBF(ARD) log_likelihood(theta), one evaluation row-sharded over the GPUs",
                       "n": n, "d": d, "kernel": "rbf_ard", "parallelism": r["parallelism"]},
            "whole_eval_tflops_equiv": r["tflops_per_gpu"] * world, "loglik_last": r["loglik_last"],
            "all_gather_bytes_per_rank_per_eval": r["all_gather_bytes_per_rank_per_eval"],
            "roofline": {"kernel": "gemm_f64_kernel<0, 0, 1, 0> (row-sharded trailing update)", "bound": "mfma",
                         "achieved": r["tflops_per_gpu"], "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": r["tflops_per_gpu"] / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                         "note": "whole-evaluation N^3/3 flops per GPU-second (collectives and solves included)"},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--npoints", dest="n", type=int, default=50000, help="data points (use --npoints under torchrun: its parser claims --n)")
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--outer-block", type=int, default=0, help="K of the trailing SYRK (0 = library default)")
    ap.add_argument("--lookahead", type=int, default=-1, help="1/0: factor the next panel on a side stream (-1 = library default)")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N>1: independent replicas (one theta stream per GPU, default) or ONE evaluation row-sharded "
                         "over the GPUs (block-cyclic rows, RCCL all-gather of panel factors; strong scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for tests that "
                                                      "put several ranks on one GPU)")
    ap.add_argument("--no-sharded", action="store_true",
                    help="N>1, replicas mode: skip the extra row-sharded measurement reported under \"sharded\"")
    ap.add_argument("--sharded-timeout", type=float, default=240.0)
    ap.add_argument("--cpu-sample-n", type=int, default=18000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        ap.error("--steps must be >= 1 and --warmup >= 0")

    import torch
    from fvgp_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("FVGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # FVGP_DEVICE: ranks sharing a GPU (tests)
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend=args.backend)

    n, d = args.n, args.d
    x, y = synth(n, d)
    if args.mode == "sharded":
        return sharded_main(args, x, y, world, rank, local, dist)
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

    vals = []
    for t in range(args.warmup):
        vals.append(H.loglik(0, xd, theta_at(t), vd, ymd, KV, alpha))
    H.set_option("profile", 1)
    prof = {"launches": 0.0, "ms": 0.0, "flops": 0.0, "potrf_ms": 0.0, "kmat_ms": 0.0, "kmat_bytes": 0.0, "tail_ms": 0.0}
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

    if rank == 0:
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")
        if os.path.exists(tfile) and n == 50000:
            # HBM-side bytes per trailing-update launch from the committed rocprofv3 --pmc passes of this same
            # command (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE); not re-measured live
            traffic = json.load(open(tfile)).get("bytes_per_launch")
        evals = args.steps * world
        ms_per_step = 1e3 * elapsed / args.steps
        syrk_tflops = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
        potrf_tflops = (args.steps * (n ** 3) / 3.0) / (prof["potrf_ms"] * 1e-3) / 1e12 if prof["potrf_ms"] > 0 else 0.0
        out = {
            "metric": "log_marginal_likelihood_evals_per_sec", "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
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
                "kernel": "gemm_f64_kernel<0, 0, 1, 0> (trailing update of the blocked Cholesky, lower tiles)",
                "bound": "mfma", "achieved": syrk_tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": syrk_tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                "launches": prof["launches"], "avg_launch_ms": prof["ms"] / max(prof["launches"], 1.0),
                "algorithmic_flops_per_launch": prof["flops"] / max(prof["launches"], 1.0),
            },
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(n, d, args.cpu_sample_n)
    else:
        out = None

    # N > 1: the same workload once more as ONE evaluation row-sharded over the ranks (the partitioning the
    # north star names), outside the timed region above, reported under "sharded".  A watchdog prints the line
    # without it if the collectives do not come back, so the headline number can never be lost to this leg.
    if world > 1 and not args.no_sharded:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(args.sharded_timeout):
                if rank == 0 and out is not None:
                    out["sharded"] = {"error": f"no result within {args.sharded_timeout:.0f} s"}
                    print(json.dumps(out), flush=True)
                os._exit(3)            # a hung collective is a failed run: the launcher must see it

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            del KV, alpha
            torch.cuda.empty_cache()
            sh = sharded_measure(args, x, y, world, rank, local, dist, steps=3, warmup=1,
                                 check_theta=theta0 * (1.0 + 0.02 * ((args.warmup + args.steps - 1) * world)),
                                 check_value=ll if rank == 0 else None)
        except Exception as e:                                  # noqa: BLE001 -- reported, not swallowed
            sh = {"error": repr(e)[:300]}
        if rank == 0:
            out["sharded"] = sh
            print(json.dumps(out), flush=True)
            out = None
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
        done.set()
        H.close()
        return
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    H.close()


if __name__ == "__main__":
    main()
