"""Slot occupancy of one launch of the resident panel kernel over time (option "chain_stamps"): how many workgroups are alive, how many of
them are still in their products (start .. "updates done"), how many are past them (waiting for / running their solve), when the leaves end.
  python tools/chain_occupancy.py N [launch_index] [bin_us] [key=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
binw = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
H = _lib.Handle(0)
for kv in sys.argv[4:]:
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
tk = (w[:, 2] >> 24).astype(np.int64)
T = (w[:, 3] - t0) / 100.0
code = w[:, 1]
start = dict(zip(tk[code == 0], T[code == 0]))
upd = dict(zip(tk[code == 10], T[code == 10]))
solved = dict(zip(tk[code == 3], T[code == 3]))
end = {}
for c in (4, 2, 9):
    for a, b in zip(tk[code == c], T[code == c]):
        end[a] = max(end.get(a, 0.0), b)
colk = dict(zip(tk[code == 10], (w[:, 2] & 255)[code == 10]))
span = T.max()
print(f"N={n} launch {which}: {len(start)} workgroups, span {span:.0f} us; leaves done at:", " ".join(f"{t:.0f}" for t in sorted(T[code == 2])))
nb = int(span // binw) + 1
alive = np.zeros(nb); prod = np.zeros(nb); post = np.zeros(nb)
def add(arr, a, b):
    if b <= a: return
    i0, i1 = int(a // binw), int(b // binw)
    for i in range(i0, min(i1, nb - 1) + 1):
        lo, hi = max(a, i * binw), min(b, (i + 1) * binw)
        if hi > lo: arr[i] += (hi - lo) / binw
for t_ in start:
    a = start[t_]; b = end.get(t_, span)
    add(alive, a, b)
    if t_ in upd:
        add(prod, a, upd[t_]); add(post, upd[t_], b)
print(" t(us)   alive  in products  past products (solve / waiting for the leaf)")
for i in range(nb):
    print(f"{i * binw:6.0f}  {alive[i]:6.1f}  {prod[i]:6.1f}  {post[i]:6.1f}")
tot = span * 512
print(f"slot-time: alive {alive.sum() * binw / tot:.2f}  products(incl. their waits) {prod.sum() * binw / tot:.2f}  past products {post.sum() * binw / tot:.2f}  empty {1 - alive.sum() * binw / tot:.2f}")
# per column: duration of the product phase per product, of the rest
byk = {}
for t_ in upd:
    if t_ in start and t_ in end:
        byk.setdefault(int(colk[t_]), []).append((upd[t_] - start[t_], end[t_] - upd[t_], start[t_], end[t_]))
print(" k  blocks  products (us, per product)  after products (us)  first start  last start  last end")
for k in sorted(byk):
    a = np.array(byk[k])
    print(f"{k:2d} {len(a):6d}  {np.median(a[:, 0]):8.1f} ({np.median(a[:, 0]) / max(k, 1):6.1f})  {np.median(a[:, 1]):8.1f}   {a[:, 2].min():9.0f}  {a[:, 2].max():9.0f}  {a[:, 3].max():9.0f}")
