"""Timeline of ONE launch of the resident panel kernel (chain.hip; diagnostic option "chain_stamps"): when the runner's leaves start and
end, when the square's block rows publish, when the rows below finish.
  python tools/chain_timeline.py N [launch_index] [key=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H = _lib.Handle(0)
for kv in sys.argv[3:]:
    H.set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
xd = H.to_device(x); npad = _lib.pad128(n)
ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
theta = np.array([1.0, 0.3, 0.3, 0.3])
H.loglik(0, xd, theta, V, ym, KV, alpha)
stamps = torch.zeros(8 + 4 * (1 << 20), dtype=torch.int64, device="cuda")
H.set_option("chain_stamps", stamps.data_ptr())
H.loglik(0, xd, theta * 1.01, V, ym, KV, alpha)
torch.cuda.synchronize()
H.set_option("chain_stamps", 0)
s = stamps.cpu().numpy()
cnt = int(s[0]); e = s[8:8 + 4 * cnt].reshape(cnt, 4)
seqs = np.unique(e[:, 0])
print("events", cnt, "launches", len(seqs))
names = {0: "start", 1: "leaf_begin", 2: "leaf_done", 3: "solver: block solved", 4: "solver: block published", 5: "diagonal block up to date", 6: "below_sum_begin",
         7: "below_sum_end", 8: "below_trsm_begin", 9: "below_trsm_end", 10: "block: updates done"}
for q in ([seqs[which]] if which >= 0 else seqs):
    w = e[e[:, 0] == q]
    w = w[np.argsort(w[:, 3])]
    t0 = w[:, 3].min()
    span = (w[:, 3].max() - t0) / 100.0
    starts = w[w[:, 1] == 0]
    print(f"launch {int(q)}: {len(starts)} workgroups, starts spread {(starts[:, 3].max() - t0) / 100.0:.1f} us, span {span:.1f} us")
    if which < 0:
        continue
    for row in w:
        code, packed, t = int(row[1]), int(row[2]), (row[3] - t0) / 100.0
        tk, r, st = packed >> 24, (packed >> 8) & 0xffff, packed & 255
        if code in (1, 2, 3, 4, 5, 10) or (code in (6, 7, 8, 9) and r == int(w[:, 2].max() >> 8) & 0xffff) or (code == 0 and tk < 12):
            print(f"  {t:9.1f} us  ticket {tk:4d} row {r:3d} step {st:2d}  {names[code]}")
