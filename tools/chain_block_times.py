"""Where a block's workgroup of the resident panel kernel spends its time in a TALL panel (option "chain_stamps"): per block column k the
time from the workgroup's start to "updates done" (its k products), from there to "solved" (the substitution) and to "published".
  python tools/chain_block_times.py N [launch_index]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
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
cnt = min(int(s[0]), 1 << 20); e = s[8:8 + 4 * cnt].reshape(cnt, 4)
seqs = np.unique(e[:, 0])
w = e[e[:, 0] == seqs[which]]
t0 = w[:, 3].min()
tk = w[:, 2] >> 24
start = {int(a): (b - t0) / 100.0 for a, b in zip(tk[w[:, 1] == 0], w[w[:, 1] == 0][:, 3])}
rec = {}
for code, packed, t in zip(w[:, 1], w[:, 2], w[:, 3]):
    if code in (10, 3, 4):
        rec.setdefault(int(packed >> 24), {})[int(code)] = ((t - t0) / 100.0, int((packed >> 8) & 0xffff), int(packed & 255))
span = (w[:, 3].max() - t0) / 100.0
print(f"launch {which}: {len(start)} workgroups, span {span:.0f} us")
leaf = sorted((r[3], t) for code, p, t in zip(w[:, 1], w[:, 2], w[:, 3]) if code == 2 for r in [((p >> 24), (p >> 8) & 0xffff, p & 255, (t - t0) / 100.0)])
print("leaves done at (us):", " ".join(f"{a:.0f}" for a, _ in leaf))
byk = {}
for ticket, r in rec.items():
    if 10 in r and 3 in r and 4 in r and ticket in start:
        k = r[10][2]
        byk.setdefault(k, []).append((r[10][0] - start[ticket], r[3][0] - r[10][0], r[4][0] - r[3][0], start[ticket]))
print(" k  blocks  start..updates done (per block column)  ..solved  ..published   started at (median)")
for k in sorted(byk):
    a = np.array(byk[k])
    print(f"{k:2d} {len(a):6d}  {np.median(a[:, 0]):8.1f} us ({np.median(a[:, 0]) / max(k, 1):6.1f})  {np.median(a[:, 1]):8.1f}  {np.median(a[:, 2]):6.1f}   {np.median(a[:, 3]):9.1f}")
