"""Does a copy-engine (SDMA) device-to-device copy keep its rate beside a saturated trailing update?  The row-sharded path's RCCL
all-gathers run copy KERNELS that queue for compute units (tools/rccl_beside_update.py: 9x slower beside the update); a gather
built on hipMemcpyAsync(..., hipMemcpyDeviceToDeviceNoCU) would not.  One GPU: the same panel-sized copies on a second stream,
alone and beside back-to-back 8192^3 fp64 products on the main stream, by copy kernel (hipMemcpyDeviceToDevice) and by copy engine.
  python tools/copy_engine_beside_update.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fvgp_amd import _lib  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
D2D, D2D_NOCU = 3, 1024

H = _lib.Handle(0)
n = 8192
A = H.to_device(np.random.default_rng(0).random((n, n)))
C = H.zeros(n, n)
side = torch.cuda.Stream()
sizes = [int(s) for s in np.linspace(48, 4, 12) * 1024 * 1024]          # doubles: 384 MB ... 32 MB, a panel gather's sizes at N = 50k
src = H.zeros(max(sizes)); dst = H.zeros(max(sizes))


def copies(kind):
    ev = []
    with torch.cuda.stream(side):
        for s in sizes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            rc = hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), s * 8, kind, side.cuda_stream)
            assert rc == 0, rc
            e1.record(side)
            ev.append((e0, e1))
    return ev


def run(kind, busy):
    torch.cuda.synchronize()
    if busy:
        for _ in range(12):                 # the handle enqueues on torch's current stream, asynchronously
            H.gemm(0, 0, 0, n, n, n, -1.0, A, A, 1.0, C)
    t0 = time.perf_counter()
    ev = copies(kind)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sum(a.elapsed_time(b) for a, b in ev)
    return ms, 8e-9 * sum(sizes) / (1e-3 * ms), wall


for kind, name in ((D2D, "copy kernel (hipMemcpyDeviceToDevice)"), (D2D_NOCU, "copy engine (hipMemcpyDeviceToDeviceNoCU)")):
    run(kind, False)
    a = run(kind, False)
    b = run(kind, True)
    print(f"{name}: {8e-9 * sum(sizes):.2f} GB in {len(sizes)} copies: alone {a[0]:.2f} ms ({a[1]:.0f} GB/s), beside the products {b[0]:.2f} ms "
          f"({b[1]:.0f} GB/s), x{b[0] / a[0]:.2f}; products + copies wall {1e3 * b[2]:.0f} ms", flush=True)
