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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--panel", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--reserve-cus", type=int, default=0)
    ap.add_argument("--chain-masked", type=int, default=0)
    args = ap.parse_args()
    import torch
    from fvgp_amd.dist import ShardedGP, HipOps

    class Emulated(ShardedGP):
        def _all_reduce(self, t):
            t.diagonal().add_(1.0e3)

        def _all_gather(self, out, inp):
            out.copy_(inp.unsqueeze(0).expand_as(out))

    rng = np.random.default_rng(20240501)
    x = rng.random((args.n, 3))
    y = np.sin(3 * x.sum(axis=1)) + 0.1 * rng.standard_normal(args.n)
    gp = Emulated(x, y, np.full(args.n, 0.01), kernel="rbf_ard", panel=args.panel, rank=args.rank, world=args.world,
                  ops=HipOps(reserve_cus=args.reserve_cus, chain_everywhere=not args.chain_masked))
    gp._into_tensor = False
    gp.keep_factor = False           # likelihood-only evaluation, as bench.py times it
    theta = np.array([1.0, 0.3, 0.3, 0.3])

    host = []

    def once():
        h0 = time.perf_counter()
        with gp.ops.stream():
            gp.assemble(theta)
            gp.factor()
        host.append(time.perf_counter() - h0)
        torch.cuda.synchronize()

    once()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        once()
    dt = (time.perf_counter() - t0) / args.steps
    flops = args.n ** 3 / 3.0
    print(f"world {args.world} rank {args.rank} n {args.n} panel {args.panel} reserve {args.reserve_cus}: {1e3 * dt:.1f} ms per evaluation "
          f"(compute-side floor), {flops / dt / 1e12 / args.world:.1f} TFLOP/s per GPU equivalent; "
          f"host enqueue {1e3 * min(host):.1f} ms")


if __name__ == "__main__":
    main()
