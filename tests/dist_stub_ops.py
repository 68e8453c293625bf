"""CPU stand-in for fvgp_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY.

Same method surface on host memory, so the partition and collective logic of the row-sharded path can run
under gloo on a machine without a GPU:
  * the sharded evaluation itself (fvgp_hip_loglik_dist) and the collective entries are the CPU twin of the C ABI
    (oracle/_cpu/libfvgp_cpu.so: the SAME driver, fvgp_amd/csrc/dist_driver.h, over host loops), bound with the
    package's own ctypes declarations;
  * the operations ShardedGP sequences itself after the factorisation (products, triangular solves, the trace pass) are
    plain torch / numpy arithmetic with the oracle's kernels.
Never imported by the package."""
import contextlib
import ctypes
import os
import subprocess

import numpy as np
import torch

from fvgp_amd import _lib
from oracle import fvgp_oracle as orc

NAMES = {0: "rbf_ard", 1: "matern32_ard", 2: "matern52_ard", 3: "rbf_iso", 4: "matern32_iso", 5: "matern52_iso"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CPU = None


def cpu_abi():
    """the CPU twin of the ABI, loaded once (built by __graft_entry__.build(); built here if missing)"""
    global _CPU
    if _CPU is None:
        path = os.environ.get("FVGP_CPU_LIB") or os.path.join(ROOT, "oracle", "_cpu", "libfvgp_cpu.so")     # (make -C oracle asan-test: the sanitizer build)
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
        _CPU = _lib.bind_dist(ctypes.CDLL(path))
        _CPU.fvgp_hip_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]
        _CPU.fvgp_hip_last_error_string.restype = ctypes.c_char_p
    return _CPU


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} (CPU twin) failed with status {rc}: {cpu_abi().fvgp_hip_last_error_string().decode()}")


class StubOps(_lib.DistCalls):
    torch = torch
    native_collectives = False

    def __init__(self):
        self._h = ctypes.c_void_p()
        _chk(cpu_abi().fvgp_hip_create(ctypes.byref(self._h), 0, None), "fvgp_hip_create")

    def stream(self):
        return contextlib.nullcontext()

    def wrap(self, ptr, count):
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_double * int(count)).from_address(int(ptr))))

    def host_sync(self):
        pass

    # -- the ABI's row-sharded entries on the CPU twin ----------------------------------------------------------------
    def dist_workspace(self, desc):
        out = (ctypes.c_int64 * 6)()
        _chk(cpu_abi().fvgp_hip_dist_workspace(ctypes.byref(desc), out), "fvgp_hip_dist_workspace")
        return list(out)

    def loglik_dist(self, desc, theta):
        t = np.ascontiguousarray(theta, dtype=np.float64)
        out = (ctypes.c_double * 3)()
        info = ctypes.c_int(0)
        _chk(cpu_abi().fvgp_hip_loglik_dist(self._h, ctypes.byref(desc), t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(t.size),
                                            out, ctypes.byref(info)), "fvgp_hip_loglik_dist")
        return out[0], out[1], out[2], info.value

    def comm_init_callbacks(self, coll, rank, nranks):
        self._coll = coll
        _chk(cpu_abi().fvgp_hip_comm_init_callbacks(self._h, ctypes.byref(coll), int(rank), int(nranks)), "fvgp_hip_comm_init_callbacks")

    def all_reduce(self, t):
        assert t.is_contiguous()
        _chk(cpu_abi().fvgp_hip_all_reduce(self._h, ctypes.c_void_p(t.data_ptr()), t.numel()), "fvgp_hip_all_reduce")

    def all_gather(self, send, recv):
        _chk(cpu_abi().fvgp_hip_all_gather(self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel()),
             "fvgp_hip_all_gather")

    def comm_profile(self):
        return {}

    def _dist_lib(self):
        return cpu_abi()

    @staticmethod
    def _dist_check(rc, what):
        _chk(rc, what)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or torch.float64)

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def kmat(self, kernel_id, x1, x2, theta, out, vdiag=None, pad=2):
        k = orc.KERNELS[NAMES[kernel_id]](x1.numpy(), x2.numpy(), np.asarray(theta))
        r = (len(x1) + 127) // 128 * 128
        c = (len(x2) + 127) // 128 * 128
        out[:r, :c] = 0.0
        out[:k.shape[0], :k.shape[1]] = torch.as_tensor(k)

    def add_matrix(self, A, B, alpha=1.0):
        A[:B.shape[0], :B.shape[1]] += alpha * B

    def sync(self):
        pass
