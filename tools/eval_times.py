"""Same-box, same-process A/B of the fused log-likelihood evaluation: wall time (best of `reps`) per size for several option sets.
   python tools/eval_times.py 8000,20000,50000 3 "" "chain_yield=2" "outer_block_big=1024,big_threshold=0"
Each quoted argument is one configuration (comma-separated key=value options of fvgp_hip_set_option; "" = defaults)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib

sizes = [int(v) for v in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
configs = sys.argv[3:] or [""]
theta = np.array([1.0, 0.3, 0.3, 0.3])
for n in sizes:
    rng = np.random.default_rng(20240501)
    x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    npad = _lib.pad128(n)
    res = []
    vals = []
    for cfg in configs:
        H = _lib.Handle(0)
        for kv in [c for c in cfg.split(",") if c]:
            k, v = kv.split("="); H.set_option(k, int(v))
        xd = H.to_device(x); ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
        dim = npad if os.environ.get("FVGP_SCRATCH_PADDED_DIM") else _lib.loglik_dim(n, 1)      # (the facade's choice: fvgp_hip_loglik_dim)
        V = H.to_device(np.full(n, 0.01)); KV = H.empty(dim, dim); alpha = H.empty(npad, 1)
        H.loglik(0, xd, theta, V, ym, KV, alpha); torch.cuda.synchronize()
        best = 1e9
        for t in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = H.loglik(0, xd, theta * (1.0 + 0.001 * t), V, ym, KV, alpha)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        vals.append(H.loglik(0, xd, theta, V, ym, KV, alpha)[0])
        res.append(best * 1e3)
        del KV, alpha; H.close(); torch.cuda.empty_cache()
    print(f"N {n}: " + "   ".join(f"[{c or 'default'}] {r:.3f} ms" for c, r in zip(configs, res)) +
          f"   (max rel diff of the values {max(abs(v - vals[0]) / abs(vals[0]) for v in vals):.1e})", flush=True)
