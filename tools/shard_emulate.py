"""Timing emulation of ONE rank of a P-rank row-sharded evaluation on a single GPU (no process group).

The collectives are replaced by local device work of the same size (all_gather: the rank's own panel rows are
copied into every chunk; all_reduce: a multiple of the identity is added so the diagonal block stays positive
definite), so the numbers are NOT a likelihood -- only the per-rank kernel / stream schedule is real.  It gives
the compute-side floor of an evaluation at world size P before xGMI latency and bandwidth are added.

    python tools/shard_emulate.py --world 8 --n 50000 --panel 1024
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def emulate(n, world=8, rank=0, panel=1024, steps=3, opts=()):
    """milliseconds per evaluation of rank `rank`'s schedule of a `world`-rank row-sharded evaluation, its collectives replaced by local
    copies of the same size (see the module text); also the host's enqueue time"""
    import torch
    from fvgp_amd import _lib
    from fvgp_amd.dist import ShardedGP, HipOps

    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3))
    y = np.sin(3 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    ops = HipOps()

    # the collectives of rank `rank` replaced by device work of the same size on the same stream: the all-gather copies the
    # rank's own rows into every chunk (the diagonal block then is not the true one, so the numbers are NOT a likelihood)
    def all_gather(ctx, send, recv, count, stream):
        s, r = ops.wrap(send, count), ops.wrap(recv, count * world)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            r.view(world, count).copy_(s.unsqueeze(0).expand(world, count))
        return 0

    def all_reduce(ctx, buf, count, stream):
        return 0

    coll = _lib.Collectives(None, _lib.ALL_GATHER_FN(all_gather), _lib.ALL_REDUCE_FN(all_reduce))
    gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel=panel, rank=rank, world=world, ops=ops, collectives=coll)
    for kv in opts:
        ops.H.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    host = []

    def once():
        h0 = time.perf_counter()
        try:
            gp.evaluate(theta, keep_factor=False)
        except np.linalg.LinAlgError:
            pass                                   # the emulated diagonal blocks need not be positive definite: only the schedule counts
        host.append(time.perf_counter() - h0)
        torch.cuda.synchronize()

    once()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    dt = (time.perf_counter() - t0) / steps
    enq = ops.H.get_profile().get("host_enqueue_ms", float("nan"))
    del gp
    torch.cuda.empty_cache()
    return 1e3 * dt, 1e3 * min(host), enq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--panel", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (repeatable)")
    args = ap.parse_args()
    ms, host_ms, enq = emulate(args.n, args.world, args.rank, args.panel, args.steps, args.opt)
    flops = args.n ** 3 / 3.0
    print(f"world {args.world} rank {args.rank} n {args.n} panel {args.panel}: {ms:.1f} ms per evaluation "
          f"(compute-side floor), {flops / (ms * 1e-3) / 1e12 / args.world:.1f} TFLOP/s per GPU equivalent; "
          f"host call {host_ms:.1f} ms, of which enqueue {enq:.1f} ms")


if __name__ == "__main__":
    main()
