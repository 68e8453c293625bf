"""Phase breakdown of the 128 x 128 leaf kernel from in-kernel s_memtime stamps (diagnostic option "leaf_stamps")."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
H = _lib.Handle(0)
if len(sys.argv) > 1: H.set_option("panel_chain", int(sys.argv[1]))
if len(sys.argv) > 2: H.set_option("leaf_tiles", int(sys.argv[2]))
rng = np.random.default_rng(0)
B = rng.standard_normal((256, 256)); M = B @ B.T + 256 * np.eye(256)
A = H.to_device(np.tril(M))
stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
H.set_option("leaf_stamps", stamps.data_ptr())
for rep in range(3):
    A.copy_(H.to_device(np.tril(M)))
    H.potrf(A, 256)
torch.cuda.synchronize()
H.set_option("leaf_stamps", 0)
s = stamps.cpu().numpy()
names = ["load (LDS-DMA) + zero upper + sync", "diag tile 0 (wave 0)", "sync"]
for p in range(7):
    names += [f"p={p} wave 0: solve tile ({p + 1},{p})", f"p={p} wave 0: update tile ({p + 1},{p + 1})", f"p={p} wave 0: factor tile {p + 1}", f"p={p} sync"]
names += ["last tile store + logdet", "tile 7 inverse + sync", "tail (block-column inverse + store, or nothing)"]
d = np.diff(s[:len(names) + 1])
for nme, c in zip(names, d):
    print(f"{nme:36s} {c:8d} cycles")
print("total", s[len(names)] - s[0], "cycles =", (s[len(names)] - s[0]) / 2.4e3, "us at 2.4 GHz")
