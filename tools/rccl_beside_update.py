"""One rank through RCCL on one GPU (ShardedGP(force_collectives=True)): what the collectives of a row-sharded evaluation cost on the
chain stream BESIDE the rank's saturated trailing update, against the same calls replayed ALONE on the idle chip
(fvgp_hip_comm_profile: calls, ms on their stream).  At one rank an all-gather moves no bytes between GPUs -- RCCL runs its copy
kernel -- so this times exactly the part a multi-rank run cannot avoid either: a collective's kernel queueing for compute units.
  python tools/rccl_beside_update.py [N] [panel] [update CUs] [rccl|ipc]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(n, panel=1024, keep_cus=256, collectives="rccl"):
    import torch
    from fvgp_amd import _lib
    from fvgp_amd.dist import ShardedGP, TILE, HipOps
    ctx = None
    if keep_cus < 256:
        # the rank's main stream (assembly, trailing updates) restricted to `keep_cus` compute units, the same number on every XCD
        # (mask bit i = compute unit i / 8 of XCD i % 8): the rest stay free for the chain stream's collectives and panel chain
        sp = _lib.create_stream(0, cu_mask=range(keep_cus))
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(sp))
        ctx.__enter__()
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    ops = HipOps(_lib.Handle(0)) if keep_cus < 256 else None
    sh = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel=panel, rank=0, world=1, force_collectives=True, collectives=collectives, ops=ops)
    th = np.array([1.0, 0.3, 0.3, 0.3])
    o = sh.ops
    sh.log_likelihood(th)
    o.set_option("profile", 1)
    o.comm_profile()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh.log_likelihood(th * 1.01)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    beside = o.comm_profile()["all_gather"]
    # the same calls alone: per panel the diagonal block (NB x w doubles at one rank) and the rows from the panel's first block down
    npad = sh.np_
    sizes = []
    for J in range(sh.npan):
        J0, Jend = sh.bnd[J], sh.bnd[J + 1]
        w = Jend - J0
        sizes.append(panel * w)
        if Jend < npad:
            sizes.append((sh.nb_max - Jend // TILE) * TILE * w)
    big = max(sizes)
    send = o.zeros(big); recv = o.zeros(big)
    with o.stream():
        for c in sizes:
            o.all_gather(send[:c], recv[:c])
        o.sync()
    o.comm_profile()
    with o.stream():
        for c in sizes:
            o.all_gather(send[:c], recv[:c])
        o.sync()
    alone = o.comm_profile()["all_gather"]
    o.set_option("profile", 0)
    if ctx is not None:
        ctx.__exit__(None, None, None)
    return {"collectives": collectives, "n": n, "panel": panel, "update_cus": keep_cus, "evaluation_ms": 1e3 * wall, "calls": beside[0], "doubles_moved": float(sum(sizes)),
            "all_gather_ms_beside_update": beside[2], "all_gather_ms_alone": alone[2], "calls_alone": alone[0],
            "slowdown": beside[2] / max(alone[2], 1e-9)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    panel = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    coll = sys.argv[4] if len(sys.argv) > 4 else "rccl"          # "rccl" | "ipc" (csrc/ipc.hip: copies between window mappings + one-wave flag kernels)
    for keep in ([int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256]):
        print(measure(n, panel, keep, coll), flush=True)
