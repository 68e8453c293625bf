"""Many evaluations of the same theta, plain and with the hand-off checksums: every run must give the first run's bits.
   python tools/chain_verify_stress.py [N] [plain reps] [verify reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
rp = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rv = int(sys.argv[3]) if len(sys.argv) > 3 else 200
H = _lib.Handle(0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); H.set_option(k, int(v))
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
npad = _lib.pad128(n)
xd = H.to_device(x); ym = H.zeros(npad, 1); ym[:n, 0] = H.to_device(y - y.mean())
V = H.to_device(np.full(n, 0.01)); KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
theta = np.array([1.0, 0.3, 0.3, 0.3])
KV.fill_(float("nan"))
ref = H.loglik(0, xd, theta, V, ym, KV, alpha)
Lref = KV[:n, :n].tril().clone()
print("first :", ref, flush=True)
for mode, reps in ((0, rp), (1, rv)):
    H.set_option("chain_verify", mode)
    H.chain_verify_counts()
    nbad = 0
    for t in range(reps):
        KV.fill_(float("nan"))
        out = H.loglik(0, xd, theta, V, ym, KV, alpha)
        L = KV[:n, :n].tril()
        diff = (L != Lref)
        nd = int(diff.sum().item())
        if nd or out != ref:
            nbad += 1
            bad, checks = H.chain_verify_counts() if mode else (0, 0)
            where = ""
            if nd:
                idx = diff.nonzero()
                br, bc = idx[:, 0] // 128, idx[:, 1] // 128
                c0 = int(bc.min()); rows_c0 = sorted(set(br[bc == c0].tolist()))
                r0 = int(idx[bc == c0][:, 0].min()); cc0 = int(idx[(bc == c0)][:, 1].min())
                mx = float((L - Lref)[diff].abs().max().item())
                rb0 = rows_c0[0]
                blk = diff[rb0 * 128:rb0 * 128 + 128, c0 * 128:c0 * 128 + 128]
                ro = sorted(set(blk.nonzero()[:, 0].tolist())); co = sorted(set(blk.nonzero()[:, 1].tolist()))
                def rng_(v):
                    out, a = [], None
                    for e in v + [None]:
                        if a is None: a = b = e
                        elif e is not None and e == b + 1: b = e
                        else:
                            out.append(f"{a}-{b}" if b != a else f"{a}"); a = b = e
                    return ",".join(out)
                print(f"   block ({rb0},{c0}): {int(blk.sum().item())} entries differ; row offsets {rng_(ro)}; col offsets {rng_(co)}")
                where = f" first differing block column {c0}: block rows {rows_c0[:10]}{'...' if len(rows_c0) > 10 else ''} ({len(rows_c0)}), first row {r0} col {cc0}; max |dL| {mx:.3e}"
            print(f"mode {mode} run {t}: {out} checksum mismatches {bad}/{checks}; differing entries {nd}{where}", flush=True)
    print(f"mode {mode}: {nbad} of {reps} runs differ", flush=True)
H.set_option("chain_verify", 0)
