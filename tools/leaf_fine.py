"""Every wave's event times in the factor loop of the 128 x 128 leaf: diagnostic build only.
   tools/build_variant.sh fine -DFVGP_LEAF_FINE && FVGP_HIP_LIB=fvgp_amd/csrc/variants/fine/libfvgp_hip.so python tools/leaf_fine.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fvgp_amd import _lib
H = _lib.Handle(0)
L = _lib.lib()
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 128)); M = B @ B.T + 128 * np.eye(128)
A = H.to_device(np.tril(M))
buf = (ctypes.c_ulong * 512)()
for rep in range(3):
    A.copy_(H.to_device(np.tril(M)))
    torch.cuda.synchronize()
    H.potrf(A, 128)
    torch.cuda.synchronize()
L.fvgp_hip_debug_fine(buf, 512)
s = np.array(buf[:], dtype=np.int64).reshape(8, 8, 8)
ev = ["start", "solved+signalled", "counter reached", "targets done", "stores done", "inverse / factor done", "past barrier"]
for p in range(7):
    t0 = s[0, p, 0]
    print(f"step {p} (cycles after wave 0's step start)")
    for w in range(8):
        row = s[w, p]
        print(f"   wave {w}: " + "  ".join(f"{ev[k].split(' ')[0]} {int(row[k] - t0) if row[k] else '-':>6}" for k in range(7)))
