"""ctypes binding of libfvgp_hip.so (include/fvgp_hip.h) -- the only door into the HIP kernels.

There is no CPU fallback: if the shared library is missing or a call fails, this module
raises.  `build()` compiles the library in-tree with hipcc for gfx950 (works without a GPU).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FVGP_HIP_LIB", os.path.join(CSRC, "libfvgp_hip.so"))     # FVGP_HIP_LIB: A/B another build of the same ABI

KERNEL_IDS = {"rbf_ard": 0, "matern32_ard": 1, "matern52_ard": 2,
              "rbf_iso": 3, "matern32_iso": 4, "matern52_iso": 5}
FULL, LOWER = 0, 1
PAD_NONE, PAD_IDENTITY, PAD_ZERO = 0, 1, 2
TILE = 128
MAX_RHS_VEC = 8

# every symbol include/fvgp_hip.h declares (tests check the library exports each of them)
SYMBOLS = [
    "fvgp_hip_version", "fvgp_hip_last_error_string", "fvgp_hip_padded_dim", "fvgp_hip_loglik_dim", "fvgp_hip_workspace_bytes", "fvgp_hip_create",
    "fvgp_hip_destroy", "fvgp_hip_sync", "fvgp_hip_stream_create", "fvgp_hip_stream_destroy", "fvgp_hip_set_option", "fvgp_hip_get_profile", "fvgp_hip_chain_verify_counts", "fvgp_hip_kmat",
    "fvgp_hip_potrf", "fvgp_hip_potrf_dev", "fvgp_hip_potrs", "fvgp_hip_logdet", "fvgp_hip_potri", "fvgp_hip_trsm_lower",
    "fvgp_hip_loglik", "fvgp_hip_loglik_grad", "fvgp_hip_grad_trace", "fvgp_hip_posterior", "fvgp_hip_gemm",
    "fvgp_hip_mfma_selftest", "fvgp_hip_mfma_peak", "fvgp_hip_symmetrize", "fvgp_hip_add_lower", "fvgp_hip_trace_dot", "fvgp_hip_colsumsq", "fvgp_hip_add_matrix", "fvgp_hip_dot", "fvgp_hip_coldot",
    "fvgp_hip_debug_tile_map", "fvgp_hip_debug_tile_table", "fvgp_hip_debug_chain_ticket", "fvgp_hip_invalidate_factor", "fvgp_hip_trsm_lower_t", "fvgp_hip_panel_trsm", "fvgp_hip_panel_potrf_dev", "fvgp_hip_syrk_rowshard",
    "fvgp_hip_grad_trace_cols", "fvgp_hip_comm_unique_id", "fvgp_hip_comm_init", "fvgp_hip_comm_init_callbacks", "fvgp_hip_comm_destroy", "fvgp_hip_ipc_window", "fvgp_hip_comm_init_ipc", "fvgp_hip_all_reduce",
    "fvgp_hip_all_gather", "fvgp_hip_comm_profile", "fvgp_hip_dist_workspace", "fvgp_hip_loglik_dist", "fvgp_hip_dist_scratch", "fvgp_hip_solve_dist",
    "fvgp_hip_posterior_dist", "fvgp_hip_grad_dist", "fvgp_hip_loglik_rows", "fvgp_hip_get_profile_ex", "fvgp_hip_comm_info", "fvgp_hip_comm_check", "fvgp_hip_posterior_prepare",
]


# ---- row-sharded entry points: structures of include/fvgp_hip.h ---------------------------------------------------
ALL_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
ALL_REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


class Collectives(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_reduce_sum", ALL_REDUCE_FN)]


class DistDesc(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("d", ctypes.c_int), ("ncol", ctypes.c_int), ("panel", ctypes.c_int64),
                ("rank", ctypes.c_int), ("nranks", ctypes.c_int), ("kernel_id", ctypes.c_int),
                ("x_all", ctypes.c_void_p), ("vdiag", ctypes.c_void_p), ("zt", ctypes.c_void_p),
                ("A", ctypes.c_void_p), ("T", ctypes.c_void_p * 2), ("recv", ctypes.c_void_p * 2), ("Dfac", ctypes.c_void_p),
                ("gather", ctypes.c_void_p), ("info_dev", ctypes.c_void_p), ("logdet_dev", ctypes.c_void_p),
                ("keep_factor", ctypes.c_int), ("force_general", ctypes.c_int), ("preassembled", ctypes.c_int)]


def bind_dist(L):
    """argument types of the row-sharded entry points on a loaded library that implements them (libfvgp_hip.so; the
    tests bind the CPU twin of the ABI the same way)"""
    c_i, c_l, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    P_d = ctypes.POINTER(ctypes.c_double)
    P_i = ctypes.POINTER(ctypes.c_int)
    if hasattr(L, "fvgp_hip_comm_unique_id"):
        L.fvgp_hip_comm_unique_id.argtypes = [c_p]
        L.fvgp_hip_comm_init.argtypes = [c_p, c_p, c_i, c_i]
        L.fvgp_hip_comm_profile.argtypes = [c_p, P_d]
    if hasattr(L, "fvgp_hip_comm_info"):
        L.fvgp_hip_comm_info.argtypes = [c_p, ctypes.POINTER(c_l)]
        L.fvgp_hip_comm_check.argtypes = [c_p]
        L.fvgp_hip_comm_info.restype = L.fvgp_hip_comm_check.restype = c_i
    L.fvgp_hip_comm_init_callbacks.argtypes = [c_p, ctypes.POINTER(Collectives), c_i, c_i]
    if hasattr(L, "fvgp_hip_ipc_window"):
        L.fvgp_hip_ipc_window.argtypes = [c_p, c_l, c_p]
        L.fvgp_hip_comm_init_ipc.argtypes = [c_p, c_p, ctypes.c_char_p, c_i, c_i]
    L.fvgp_hip_comm_destroy.argtypes = [c_p]
    L.fvgp_hip_all_reduce.argtypes = [c_p, c_p, c_l]
    L.fvgp_hip_all_gather.argtypes = [c_p, c_p, c_p, c_l]
    L.fvgp_hip_dist_workspace.argtypes = [ctypes.POINTER(DistDesc), ctypes.POINTER(c_l)]
    L.fvgp_hip_loglik_dist.argtypes = [c_p, ctypes.POINTER(DistDesc), P_d, c_i, P_d, P_i]
    L.fvgp_hip_dist_scratch.argtypes = [ctypes.POINTER(DistDesc), c_i, c_l, c_l]
    L.fvgp_hip_dist_scratch.restype = c_l
    L.fvgp_hip_solve_dist.argtypes = [c_p, ctypes.POINTER(DistDesc), c_p, c_p]
    L.fvgp_hip_posterior_dist.argtypes = [c_p, ctypes.POINTER(DistDesc), P_d, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p]
    L.fvgp_hip_grad_dist.argtypes = [c_p, ctypes.POINTER(DistDesc), P_d, c_i, c_p, c_i, c_l, P_d, c_p, c_p]
    for s in ("fvgp_hip_solve_dist", "fvgp_hip_posterior_dist", "fvgp_hip_grad_dist", "fvgp_hip_comm_unique_id", "fvgp_hip_comm_init", "fvgp_hip_comm_profile", "fvgp_hip_comm_init_callbacks", "fvgp_hip_comm_destroy",
              "fvgp_hip_all_reduce", "fvgp_hip_all_gather", "fvgp_hip_dist_workspace", "fvgp_hip_loglik_dist"):
        if hasattr(L, s):
            getattr(L, s).restype = c_i
    return L


class DistCalls:
    """The row-sharded entry points that follow a kept factorisation (include/fvgp_hip.h: fvgp_hip_solve_dist, _posterior_dist,
    _grad_dist), as methods over `self._dist_lib()` (the loaded library) and `self._h` (its handle) with `self._dist_check` as the
    status check: the product's Handle binds libfvgp_hip.so, the CPU tests' stand-in binds the CPU twin of the ABI the same way.
    Tensors are anything with data_ptr()."""

    def dist_scratch(self, desc, what, npred=0, slab=TILE):
        n = self._dist_lib().fvgp_hip_dist_scratch(ctypes.byref(desc), int(what), int(npred), int(slab))
        if n < 0:
            raise ValueError("fvgp_hip_dist_scratch: bad arguments")
        return int(n)

    def solve_dist(self, desc, alpha_out, ws):
        self._dist_check(self._dist_lib().fvgp_hip_solve_dist(self._h, ctypes.byref(desc), _ptr(alpha_out), _ptr(ws)), "fvgp_hip_solve_dist")

    def posterior_dist(self, desc, theta, xpred, npred, k_pre, kk_pre, alpha, mean_out, S_out, ws):
        t, tp, nt = _theta(theta)
        self._dist_check(self._dist_lib().fvgp_hip_posterior_dist(self._h, ctypes.byref(desc), tp, nt, _ptr(xpred), int(npred), _ptr(k_pre), _ptr(kk_pre),
                                                                  _ptr(alpha), _ptr(mean_out), _ptr(S_out), _ptr(ws)), "fvgp_hip_posterior_dist")

    def grad_dist(self, desc, theta, alpha, component, slab, diag_out, ws):
        t, tp, nt = _theta(theta)
        g = (ctypes.c_double * nt)()
        self._dist_check(self._dist_lib().fvgp_hip_grad_dist(self._h, ctypes.byref(desc), tp, nt, _ptr(alpha), int(component), int(slab), g,
                                                             _ptr(diag_out), _ptr(ws)), "fvgp_hip_grad_dist")
        return np.array(list(g))


class HipExtensionError(RuntimeError):
    """The native library is missing, failed to load, or a call into it failed."""


def build(force=False, verbose=False):
    """Compile fvgp_amd/csrc/*.hip into libfvgp_hip.so (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise HipExtensionError("building libfvgp_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


_lib = None


def pad128(n):
    return (int(n) + TILE - 1) // TILE * TILE


def loglik_dim(n, ncol=1):
    """rows and columns of the square scratch the fused evaluation wants (fvgp_hip_loglik_dim): padded_dim(n), or 128 more when n
    leaves fewer than ncol padding rows for the appended (y-m)^T"""
    n, ncol = int(n), int(ncol)
    return pad128(n) if pad128(n) - n >= ncol else pad128(n + ncol)


def lib():
    """Load (once) and return the ctypes library with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found. The HIP extension is the product path and has no fallback; "
            f"build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC}`.")
    # torch first: libfvgp_hip.so must bind to the HIP runtime torch has loaded (its own copy), not bring /opt/rocm's in
    # beside it -- two runtimes in one process, and the one loaded first sees "no ROCm-capable device" once the other owns it
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipExtensionError(f"cannot load {LIB_PATH}: {e}") from e
    c_i, c_l, c_d, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    P_d = ctypes.POINTER(ctypes.c_double)
    P_i = ctypes.POINTER(ctypes.c_int)
    L.fvgp_hip_version.restype = c_i
    L.fvgp_hip_last_error_string.restype = ctypes.c_char_p
    L.fvgp_hip_padded_dim.restype = c_l
    L.fvgp_hip_padded_dim.argtypes = [c_l]
    L.fvgp_hip_loglik_dim.argtypes = [c_l, c_i]
    L.fvgp_hip_loglik_dim.restype = c_l
    L.fvgp_hip_workspace_bytes.argtypes = [c_l, c_l]
    L.fvgp_hip_workspace_bytes.restype = c_l
    L.fvgp_hip_create.argtypes = [ctypes.POINTER(c_p), c_i, c_p]
    L.fvgp_hip_destroy.argtypes = [c_p]
    L.fvgp_hip_sync.argtypes = [c_p]
    L.fvgp_hip_stream_create.argtypes = [ctypes.POINTER(c_p), c_i, c_i, ctypes.POINTER(ctypes.c_uint32), c_i]
    L.fvgp_hip_stream_destroy.argtypes = [c_p]
    L.fvgp_hip_set_option.argtypes = [c_p, ctypes.c_char_p, c_l]
    L.fvgp_hip_get_profile.argtypes = [c_p, P_d]
    L.fvgp_hip_chain_verify_counts.argtypes = [c_p, ctypes.POINTER(ctypes.c_int64)]
    L.fvgp_hip_invalidate_factor.argtypes = [c_p]
    L.fvgp_hip_kmat.argtypes = [c_p, c_i, c_p, c_l, c_p, c_l, c_i, P_d, c_i, c_p, c_p, c_l, c_i, c_i]
    L.fvgp_hip_potrf.argtypes = [c_p, c_p, c_l, c_l, P_i]
    L.fvgp_hip_potrf_dev.argtypes = [c_p, c_p, c_l, c_l, c_l, c_p, c_p]
    L.fvgp_hip_potrs.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l, c_l]
    L.fvgp_hip_trsm_lower.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l, c_l]
    L.fvgp_hip_logdet.argtypes = [c_p, c_p, c_l, c_l, P_d]
    L.fvgp_hip_potri.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l]
    L.fvgp_hip_loglik.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_p, c_i, c_p, c_l, c_p, P_d, P_i]
    L.fvgp_hip_loglik_rows.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_p, c_i, c_p, c_l, c_l, c_p, P_d, P_i]
    L.fvgp_hip_get_profile_ex.argtypes = [c_p, P_d]
    L.fvgp_hip_loglik_grad.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_i, c_i, c_p, c_l, c_p, c_l, P_d]
    L.fvgp_hip_grad_trace.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_l, c_p, c_l, c_p, P_d]
    L.fvgp_hip_posterior.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_l, c_p, c_i, c_p, c_l,
                                     c_p, c_l, c_p, c_p, c_p, c_l]
    L.fvgp_hip_posterior_prepare.argtypes = [c_p, c_p, c_l, c_l]
    L.fvgp_hip_gemm.argtypes = [c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_d, c_p, c_l, c_p, c_l, c_d, c_p, c_l]
    L.fvgp_hip_mfma_selftest.argtypes = [c_p, c_p, c_p, c_p]
    L.fvgp_hip_symmetrize.argtypes = [c_p, c_p, c_l, c_l]
    L.fvgp_hip_add_lower.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l, c_d]
    L.fvgp_hip_trace_dot.argtypes = [c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_l, P_d]
    L.fvgp_hip_colsumsq.argtypes = [c_p, c_p, c_l, c_l, c_l, c_p]
    L.fvgp_hip_add_matrix.argtypes = [c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_d]
    L.fvgp_hip_dot.argtypes = [c_p, c_p, c_l, c_p, c_l, c_l, c_i, P_d]
    L.fvgp_hip_coldot.argtypes = [c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_p]
    L.fvgp_hip_mfma_peak.argtypes = [c_p, c_p, c_i, c_i]
    L.fvgp_hip_trsm_lower_t.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l, c_l]
    L.fvgp_hip_panel_trsm.argtypes = [c_p, c_p, c_l, c_l, c_p, c_l, c_l]
    L.fvgp_hip_panel_potrf_dev.argtypes = [c_p, c_p, c_l, c_l, c_l, c_l, c_p, c_p]
    L.fvgp_hip_syrk_rowshard.argtypes = [c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i]
    L.fvgp_hip_debug_tile_map.argtypes = [c_i, c_i, c_i, c_i, c_i, P_i, P_i, c_l]
    L.fvgp_hip_debug_tile_map.restype = c_l
    L.fvgp_hip_debug_chain_ticket.argtypes = [c_i, c_i, c_i, c_i, P_i]
    L.fvgp_hip_debug_chain_ticket.restype = c_i
    L.fvgp_hip_debug_tile_table.argtypes = [c_i, c_i, c_i, c_i, c_i, P_i, c_l]
    L.fvgp_hip_debug_tile_table.restype = c_l
    L.fvgp_hip_grad_trace_cols.argtypes = [c_p, c_i, c_p, c_l, c_i, P_d, c_i, c_p, c_l, c_l, c_l, c_p, c_l, c_p, P_d]
    bind_dist(L)
    for s in SYMBOLS:
        if s not in ("fvgp_hip_last_error_string", "fvgp_hip_padded_dim", "fvgp_hip_debug_tile_map", "fvgp_hip_debug_tile_table", "fvgp_hip_debug_chain_ticket", "fvgp_hip_workspace_bytes", "fvgp_hip_dist_scratch"):
            getattr(L, s).restype = c_i
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        msg = lib().fvgp_hip_last_error_string().decode(errors="replace")
        raise HipExtensionError(f"{what} failed with status {rc}" + (f": {msg}" if msg else ""))


def _theta(theta):
    t = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
    return t, t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(t.size)


def _ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def comm_unique_id():
    """128 bytes naming a new RCCL communicator (ncclGetUniqueId): rank 0 calls it and hands the bytes to every rank"""
    buf = ctypes.create_string_buffer(128)
    _check(lib().fvgp_hip_comm_unique_id(buf), "fvgp_hip_comm_unique_id")
    return bytes(buf.raw)


def create_stream(device, cu_mask=None, high_priority=False):
    """hipStream_t (as int) from fvgp_hip_stream_create: restricted to the CUs in `cu_mask` (iterable of CU
    indices) or an ordinary non-blocking stream."""
    out = ctypes.c_void_p()
    if cu_mask is None:
        _check(lib().fvgp_hip_stream_create(ctypes.byref(out), int(device), int(high_priority), None, 0), "fvgp_hip_stream_create")
    else:
        words = [0] * 8
        for cu in cu_mask:
            words[cu // 32] |= 1 << (cu % 32)
        arr = (ctypes.c_uint32 * 8)(*words)
        _check(lib().fvgp_hip_stream_create(ctypes.byref(out), int(device), 0, arr, 8), "fvgp_hip_stream_create")
    return out.value


def destroy_stream(stream):
    _check(lib().fvgp_hip_stream_destroy(ctypes.c_void_p(stream)), "fvgp_hip_stream_destroy")


class Handle(DistCalls):
    """One device + one stream.  Thin, argument-for-argument wrapper of the C ABI; tensors are
    torch CUDA fp64 tensors used purely as device-memory containers."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise HipExtensionError("no HIP device visible (torch.cuda.is_available() is False); "
                                    "the native path has no CPU fallback")
        self.torch = torch
        self.device = int(device)
        self._h = ctypes.c_void_p()
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fvgp_hip_create(ctypes.byref(self._h), self.device, ctypes.c_void_p(stream)), "fvgp_hip_create")
        for key in ("schedule", "lookahead", "outer_block", "outer_block_big", "big_threshold", "inner_block", "small_tile_max", "small_tile_max_update", "tile_tables", "block_inverses", "k128_kernels", "leaf_tiles", "leaf_tiles_rows", "panel_recursive", "potri_kminor", "leaf_yield", "chain_yield", "lookahead_min", "posterior_halves", "posterior_block", "outer_block_small", "small_threshold", "panel_chain", "panel_chain_min", "cols_split", "cols_split_rows", "bwd_sweep", "fwd_sweep", "chain_verify", "chain_wide", "wide_block", "wide_block_big", "wide_threshold", "wide_inner", "wide_inner_rows", "chain_sleep_rows", "chain_single_rows", "chain_ahead"):        # tuning overrides, e.g. FVGP_OUTER_BLOCK=512
            val = os.environ.get("FVGP_" + key.upper())
            if val is not None:
                self.set_option(key, int(val))

    def close(self):
        if self._h:
            lib().fvgp_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------
    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=f"cuda:{self.device}")

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=f"cuda:{self.device}")

    def to_device(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=f"cuda:{self.device}")

    def to_host(self, t):
        """Device tensor (a strided view is fine) -> numpy array.  Results up to 1 GB (a posterior covariance at 1000 points is
        8 MB, at 4000 points 128 MB) come back through torch's cached PINNED host memory, which the returned array keeps alive:
        one DMA at the link's rate instead of a staged copy into freshly mapped pages (about 0.4 ms of 8.8 at N=20k, P=1000; 128 MB:
        3 ms instead of 15).  Anything larger takes the ordinary pageable path so that kept results cannot pin host memory
        without bound."""
        if t.numel() * t.element_size() > (1 << 30) or t.numel() == 0:
            return t.cpu().numpy()
        out = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        out.copy_(t, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return out.numpy()

    def sync(self):
        _check(lib().fvgp_hip_sync(self._h), "fvgp_hip_sync")

    def set_option(self, key, value):
        _check(lib().fvgp_hip_set_option(self._h, key.encode(), int(value)), "fvgp_hip_set_option")

    def invalidate_factor(self):
        """Forget the cached diagonal-block inverses: call after writing a factor buffer by anything but potrf."""
        _check(lib().fvgp_hip_invalidate_factor(self._h), "fvgp_hip_invalidate_factor")

    def get_profile(self):
        out = (ctypes.c_double * 16)()
        _check(lib().fvgp_hip_get_profile_ex(self._h, out), "fvgp_hip_get_profile_ex")
        return {"launches": out[0], "ms": out[1], "flops": out[2], "potrf_ms": out[3],
                "kmat_ms": out[4], "kmat_bytes": out[5], "tail_ms": out[6], "host_enqueue_ms": out[7], "bytes": out[8]}

    def chain_verify_counts(self):
        """(mismatches, comparisons) of the resident panel kernel's hand-off checksums since the last call (option "chain_verify")"""
        out = (ctypes.c_int64 * 2)()
        _check(lib().fvgp_hip_chain_verify_counts(self._h, out), "fvgp_hip_chain_verify_counts")
        return int(out[0]), int(out[1])

    # -- ABI calls -----------------------------------------------------------------------------
    def kmat(self, kernel_id, x1, x2, theta, K, vdiag=None, uplo=FULL, pad=PAD_NONE):
        t, tp, nt = _theta(theta)
        _check(lib().fvgp_hip_kmat(self._h, int(kernel_id), _ptr(x1), x1.shape[0], _ptr(x2), x2.shape[0],
                                   x1.shape[1], tp, nt, _ptr(vdiag), _ptr(K), K.stride(0), int(uplo), int(pad)),
               "fvgp_hip_kmat")

    def potrf(self, A, n):
        info = ctypes.c_int(0)
        _check(lib().fvgp_hip_potrf(self._h, _ptr(A), int(n), A.stride(0), ctypes.byref(info)), "fvgp_hip_potrf")
        return info.value

    def potrs(self, L, n, B, nrhs):
        _check(lib().fvgp_hip_potrs(self._h, _ptr(L), int(n), L.stride(0), _ptr(B), int(nrhs), B.stride(0)),
               "fvgp_hip_potrs")

    def trsm_lower(self, L, n, B, nrhs):
        _check(lib().fvgp_hip_trsm_lower(self._h, _ptr(L), int(n), L.stride(0), _ptr(B), int(nrhs), B.stride(0)),
               "fvgp_hip_trsm_lower")

    def trsm_lower_t(self, L, n, B, nrhs):
        _check(lib().fvgp_hip_trsm_lower_t(self._h, _ptr(L), int(n), L.stride(0), _ptr(B), int(nrhs), B.stride(0)),
               "fvgp_hip_trsm_lower_t")

    def panel_trsm(self, D, nd, P, rows):
        _check(lib().fvgp_hip_panel_trsm(self._h, _ptr(D), int(nd), D.stride(0), _ptr(P), int(rows), P.stride(0)),
               "fvgp_hip_panel_trsm")

    def syrk_rowshard(self, M, N, K, A, B, C, scale, off, b_ranks=1, b_blocks=0, b_off=0):
        _check(lib().fvgp_hip_syrk_rowshard(self._h, int(M), int(N), int(K), _ptr(A), A.stride(0), _ptr(B), B.stride(0),
                                            _ptr(C), C.stride(0), int(scale), int(off), int(b_ranks), int(b_blocks),
                                            int(b_off)), "fvgp_hip_syrk_rowshard")

    def panel_potrf_dev(self, T, w, rows, n_valid, info_dev, logdet_dev):
        """Enqueue-only factorisation of a tall panel (diagonal block on top, this rank's rows below)."""
        _check(lib().fvgp_hip_panel_potrf_dev(self._h, _ptr(T), int(w), int(rows), T.stride(0), int(n_valid), _ptr(info_dev),
                                              _ptr(logdet_dev) if logdet_dev is not None else None), "fvgp_hip_panel_potrf_dev")

    def potrf_dev(self, A, n, n_logdet, info_dev, logdet_dev):
        """Enqueue-only potrf: info (int32 tensor) and log-det (float64 tensor) stay on the device."""
        _check(lib().fvgp_hip_potrf_dev(self._h, _ptr(A), int(n), A.stride(0), int(n_logdet), _ptr(info_dev),
                                        _ptr(logdet_dev) if logdet_dev is not None else None), "fvgp_hip_potrf_dev")

    def logdet(self, L, n):
        out = ctypes.c_double(0.0)
        _check(lib().fvgp_hip_logdet(self._h, _ptr(L), int(n), L.stride(0), ctypes.byref(out)), "fvgp_hip_logdet")
        return out.value

    def potri(self, L, n, work):
        _check(lib().fvgp_hip_potri(self._h, _ptr(L), int(n), L.stride(0), _ptr(work), work.stride(0)), "fvgp_hip_potri")

    def loglik(self, kernel_id, x, theta, vdiag, ymean, KV, alpha):
        """fvgp_hip_loglik_rows: the scratch's ROWS are passed as the tensor has them (KV.shape[0]), never inferred from its stride --
        the forward solve is fused for every n when KV is a square of loglik_dim(n, ncol)"""
        t, tp, nt = _theta(theta)
        out = (ctypes.c_double * 3)()
        info = ctypes.c_int(0)
        n, d = x.shape
        if KV.dim() != 2 or KV.stride(1) != 1 or KV.shape[1] < pad128(n) or KV.shape[0] < pad128(n):
            raise ValueError(f"loglik: KV must be a row-major 2-d tensor of at least {pad128(n)} x {pad128(n)}, got {tuple(KV.shape)} strides {KV.stride()}")
        # the extra block row is only claimed when the tensor owns those rows AND columns (a column slice with a wide stride does not)
        rows = KV.shape[0] if KV.shape[1] >= loglik_dim(n, ymean.shape[1]) else pad128(n)
        _check(lib().fvgp_hip_loglik_rows(self._h, int(kernel_id), _ptr(x), n, d, tp, nt, _ptr(vdiag), _ptr(ymean),
                                          ymean.shape[1], _ptr(KV), rows, KV.stride(0), _ptr(alpha), out, ctypes.byref(info)),
               "fvgp_hip_loglik_rows")
        return out[0], out[1], out[2], info.value

    def loglik_grad(self, kernel_id, x, theta, alpha, ncol, component, KV, work):
        t, tp, nt = _theta(theta)
        g = (ctypes.c_double * nt)()
        n, d = x.shape
        _check(lib().fvgp_hip_loglik_grad(self._h, int(kernel_id), _ptr(x), n, d, tp, nt, _ptr(alpha), int(ncol),
                                          int(component), _ptr(KV), KV.stride(0), _ptr(work), work.stride(0), g),
               "fvgp_hip_loglik_grad")
        return np.array(g[:], dtype=np.float64)

    def grad_trace(self, kernel_id, x, theta, W, b, partial):
        """1/2 sum_jk (W_jk - b_j b_k) dK_jk/dtheta_i over the symmetric W (lower triangle read); b a 1-d view or None."""
        t, tp, nt = _theta(theta)
        g = (ctypes.c_double * nt)()
        n, d = x.shape
        _check(lib().fvgp_hip_grad_trace(self._h, int(kernel_id), _ptr(x), n, d, tp, nt, _ptr(W), W.stride(0),
                                         _ptr(b), 1 if b is None else b.stride(0), _ptr(partial), g), "fvgp_hip_grad_trace")
        return np.array(g[:], dtype=np.float64)

    def grad_trace_cols(self, kernel_id, x, theta, W, col0, ncols, b, partial):
        """the same pass over the slab W (n, >= ncols) of columns [col0, col0 + ncols) of the symmetric matrix"""
        t, tp, nt = _theta(theta)
        g = (ctypes.c_double * nt)()
        n, d = x.shape
        _check(lib().fvgp_hip_grad_trace_cols(self._h, int(kernel_id), _ptr(x), n, d, tp, nt, _ptr(W), W.stride(0), int(col0), int(ncols),
                                              _ptr(b), 1 if b is None else b.stride(0), _ptr(partial), g), "fvgp_hip_grad_trace_cols")
        return np.array(g[:], dtype=np.float64)

    # -- row-sharded evaluation (include/fvgp_hip.h: fvgp_hip_comm_* / fvgp_hip_loglik_dist) ---------------------------
    def comm_init(self, unique_id, rank, nranks):
        """bind RCCL: unique_id = the 128 bytes of comm_unique_id() on rank 0, handed to every rank by the caller"""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _check(lib().fvgp_hip_comm_init(self._h, buf, int(rank), int(nranks)), "fvgp_hip_comm_init")

    def comm_init_callbacks(self, coll, rank, nranks):
        self._coll = coll                                          # the callbacks must outlive the handle's use of them
        _check(lib().fvgp_hip_comm_init_callbacks(self._h, ctypes.byref(coll), int(rank), int(nranks)), "fvgp_hip_comm_init_callbacks")

    def ipc_window(self, window_bytes):
        """allocate this rank's window of the direct (IPC) collectives; returns its 64-byte handle"""
        buf = ctypes.create_string_buffer(64)
        _check(lib().fvgp_hip_ipc_window(self._h, int(window_bytes), buf), "fvgp_hip_ipc_window")
        return buf.raw

    def comm_init_ipc(self, all_handles, shm_name, rank, nranks):
        """bind the direct collectives: all_handles = the ranks' 64-byte window handles in rank order"""
        blob = ctypes.create_string_buffer(b"".join(bytes(hd) for hd in all_handles), 64 * int(nranks))
        _check(lib().fvgp_hip_comm_init_ipc(self._h, blob, shm_name.encode(), int(rank), int(nranks)), "fvgp_hip_comm_init_ipc")

    def comm_destroy(self):
        _check(lib().fvgp_hip_comm_destroy(self._h), "fvgp_hip_comm_destroy")

    def all_reduce(self, t):
        _check(lib().fvgp_hip_all_reduce(self._h, _ptr(t), t.numel()), "fvgp_hip_all_reduce")

    def all_gather(self, send, recv):
        _check(lib().fvgp_hip_all_gather(self._h, _ptr(send), _ptr(recv), send.numel()), "fvgp_hip_all_gather")

    def _dist_lib(self):
        return lib()

    @staticmethod
    def _dist_check(rc, what):
        _check(rc, what)

    def comm_info(self):
        """the handle's communicator as it reports itself (fvgp_hip_comm_info)"""
        out = (ctypes.c_int64 * 8)()
        _check(lib().fvgp_hip_comm_info(self._h, out), "fvgp_hip_comm_info")
        info = {"kind": {0: "none", 1: "rccl", 2: "ipc", 3: "callbacks"}[int(out[0])], "nranks_bound": int(out[1]), "rank_bound": int(out[2])}
        if out[0] == 1:
            info.update(nccl_comm_count=int(out[3]), nccl_comm_user_rank=int(out[4]), nccl_comm_cu_device=int(out[5]), rccl_version=int(out[6]))
        return info

    def comm_check(self):
        _check(lib().fvgp_hip_comm_check(self._h), "fvgp_hip_comm_check")

    def comm_profile(self):
        out = (ctypes.c_double * 6)()
        _check(lib().fvgp_hip_comm_profile(self._h, out), "fvgp_hip_comm_profile")
        return {"all_gather": (int(out[0]), out[2], out[4]), "all_reduce": (int(out[1]), out[3], out[5])}

    def dist_workspace(self, desc):
        out = (ctypes.c_int64 * 6)()
        _check(lib().fvgp_hip_dist_workspace(ctypes.byref(desc), out), "fvgp_hip_dist_workspace")
        return list(out)

    def loglik_dist(self, desc, theta):
        t, tp, nt = _theta(theta)
        out = (ctypes.c_double * 3)()
        info = ctypes.c_int(0)
        _check(lib().fvgp_hip_loglik_dist(self._h, ctypes.byref(desc), tp, nt, out, ctypes.byref(info)), "fvgp_hip_loglik_dist")
        return out[0], out[1], out[2], info.value

    def posterior(self, kernel_id, x, theta, L, alpha, ncol, xpred, kx, mean_out=None, var_out=None, S_out=None):
        t, tp, nt = _theta(theta)
        n, d = x.shape
        _check(lib().fvgp_hip_posterior(self._h, int(kernel_id), _ptr(x), n, d, tp, nt, _ptr(L), L.stride(0),
                                        _ptr(alpha), int(ncol), _ptr(xpred), xpred.shape[0], _ptr(kx), kx.stride(0),
                                        _ptr(mean_out), _ptr(var_out), _ptr(S_out),
                                        0 if S_out is None else S_out.stride(0)), "fvgp_hip_posterior")

    def posterior_prepare(self, L, n):
        """enqueue the inverted diagonal blocks the posterior's sweep substitutes with (fvgp_hip_posterior_prepare)"""
        _check(lib().fvgp_hip_posterior_prepare(self._h, _ptr(L), int(n), L.stride(0)), "fvgp_hip_posterior_prepare")

    def gemm(self, a_kmajor, b_nmajor, lower, M, N, K, alpha, A, B, beta, C):
        _check(lib().fvgp_hip_gemm(self._h, int(a_kmajor), int(b_nmajor), int(lower), M, N, K, float(alpha),
                                   _ptr(A), A.stride(0), _ptr(B), B.stride(0), float(beta), _ptr(C), C.stride(0)),
               "fvgp_hip_gemm")

    def mfma_selftest(self, A, B, D):
        _check(lib().fvgp_hip_mfma_selftest(self._h, _ptr(A), _ptr(B), _ptr(D)), "fvgp_hip_mfma_selftest")

    def mfma_peak(self, out, blocks, iters):
        _check(lib().fvgp_hip_mfma_peak(self._h, _ptr(out), int(blocks), int(iters)), "fvgp_hip_mfma_peak")

    def add_lower(self, A, n, B, alpha=1.0):
        """A[:n, :n] += alpha * B[:n, :n] on the lower triangle (K + V with a matrix-valued noise model)."""
        _check(lib().fvgp_hip_add_lower(self._h, _ptr(A), int(n), A.stride(0), _ptr(B), B.stride(0), float(alpha)), "fvgp_hip_add_lower")

    def trace_dot(self, W, D, b, n):
        """sum_ij (W_ij - b_i b_j) D_ij over the full n x n arrays (W symmetric, both triangles valid); b a 1-d view or None."""
        out = ctypes.c_double(0.0)
        _check(lib().fvgp_hip_trace_dot(self._h, _ptr(W), W.stride(0), _ptr(D), D.stride(0), _ptr(b),
                                        1 if b is None else b.stride(0), int(n), ctypes.byref(out)), "fvgp_hip_trace_dot")
        return out.value

    def add_matrix(self, A, B, alpha=1.0):
        """A += alpha * B on the rectangle of B's shape"""
        _check(lib().fvgp_hip_add_matrix(self._h, _ptr(A), A.stride(0), _ptr(B), B.stride(0), B.shape[0], B.shape[1], float(alpha)),
               "fvgp_hip_add_matrix")

    def dot(self, a, b, n):
        """sum of the elementwise products of the first n rows of two (rows, c) arrays"""
        out = ctypes.c_double(0.0)
        _check(lib().fvgp_hip_dot(self._h, _ptr(a), a.stride(0), _ptr(b), b.stride(0), int(n), a.shape[1], ctypes.byref(out)), "fvgp_hip_dot")
        return out.value

    def coldot(self, A, B, rows, cols, out):
        _check(lib().fvgp_hip_coldot(self._h, _ptr(A), A.stride(0), _ptr(B), B.stride(0), int(rows), int(cols), _ptr(out)), "fvgp_hip_coldot")

    def colsumsq(self, V, out):
        """out[p] = sum_i V[i][p]^2 over the rows of V"""
        _check(lib().fvgp_hip_colsumsq(self._h, _ptr(V), V.shape[0], V.stride(0), V.shape[1], _ptr(out)), "fvgp_hip_colsumsq")

    def symmetrize(self, A, n):
        _check(lib().fvgp_hip_symmetrize(self._h, _ptr(A), int(n), A.stride(0)), "fvgp_hip_symmetrize")
