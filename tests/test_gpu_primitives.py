"""HIP kernels through the C ABI (fvgp_amd/_lib.py -> libfvgp_hip.so) against numpy/scipy and the
oracle, on sizes the CPU finishes in seconds.  Mirrors tests/test_fvgp.py:44-182 of the reference
(GPU vs CPU linalg), with its rtol 1e-5 tightened to the fp64 bars of SURVEY 8c."""
import json
import os

import numpy as np
import pytest
import scipy.linalg as sla

from conftest import load_golden, synth
from oracle import fvgp_oracle as orc

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    from fvgp_amd import _lib
    h = _lib.Handle(0)
    yield h
    h.close()


def _padded(H, a, rows_pad=None, cols_pad=None, fill=0.0):
    """host (r,c) array -> device tensor padded to multiples of 128 (zeros)."""
    from fvgp_amd._lib import pad128
    r, c = a.shape
    rp = rows_pad or pad128(r)
    cp = cols_pad or pad128(c)
    buf = np.full((rp, cp), fill)
    buf[:r, :c] = a
    return H.to_device(buf)


def test_mfma_lane_maps(H):
    """A = 16x4, B = 4x16 with distinct integer entries: exact product, asymmetric operands."""
    rng = np.random.default_rng(1)
    A = rng.integers(-50, 50, (16, 4)).astype(float)
    B = rng.integers(-50, 50, (4, 16)).astype(float)
    D = H.zeros(16, 16)
    H.mfma_selftest(H.to_device(A), H.to_device(B), D)
    H.sync()
    assert np.array_equal(D.cpu().numpy(), A @ B)


@pytest.mark.parametrize("akm", [0, 1])
@pytest.mark.parametrize("bnm", [0, 1])
@pytest.mark.parametrize("lower", [0, 1])
def test_gemm_layouts(H, akm, bnm, lower):
    rng = np.random.default_rng(10 + akm * 4 + bnm * 2 + lower)
    M, N, K = 384, 256 if not lower else 384, 176
    opA = rng.standard_normal((M, K))
    opB = rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    A = H.to_device(opA.T.copy() if akm else opA)
    B = H.to_device(opB if bnm else opB.T.copy())
    C = H.to_device(C0)
    H.gemm(akm, bnm, lower, M, N, K, -0.75, A, B, 1.25, C)
    H.sync()
    ref = -0.75 * opA @ opB + 1.25 * C0
    got = C.cpu().numpy()
    scale = np.abs(opA) @ np.abs(opB)
    if lower:
        for ti in range(M // 128):
            for tj in range(N // 128):
                sl = (slice(ti * 128, ti * 128 + 128), slice(tj * 128, tj * 128 + 128))
                if tj <= ti:
                    assert np.max(np.abs(got[sl] - ref[sl]) / scale[sl]) < 8 * EPS
                else:
                    assert np.array_equal(got[sl], C0[sl]), "tile above the diagonal must be untouched"
    else:
        assert np.max(np.abs(got - ref) / scale) < 8 * EPS


@pytest.mark.parametrize("n", [100, 128, 300, 3333, 8200, 36000])
def test_backward_sweep_in_one_launch_matches_the_step_kernels(H, n):
    """fvgp_hip_potrs with one right-hand side: the single-launch backward sweep (workgroups hand x_k to each other through
    tagged 16-byte granules) must give the bits of the per-block launches -- every sum is formed in the same order, so a stale
    or torn hand-off would show.  36000 rows = 282 workgroups, more than the 256 that are resident at once."""
    from fvgp_amd._lib import pad128
    import torch
    rng = np.random.default_rng(n)
    npad = pad128(n)
    A = H.empty(npad, npad)
    A.zero_()
    # a well-conditioned lower factor without a factorisation: unit-ish diagonal, small entries below
    g = torch.Generator(device=A.device); g.manual_seed(n)
    blk = 4096
    for r0 in range(0, npad, blk):
        r1 = min(npad, r0 + blk)
        A[r0:r1, :r1] = 0.02 * torch.randn(r1 - r0, r1, dtype=torch.float64, device=A.device, generator=g) / np.sqrt(npad)
    A.diagonal().copy_(1.0 + torch.rand(npad, dtype=torch.float64, device=A.device, generator=g))
    A[n:, :] = 0.0
    A.diagonal()[n:] = 1.0
    H.invalidate_factor()
    rhs = rng.standard_normal((n, 1))
    out = {}
    for mode in (0, 1, 1):
        H.set_option("bwd_sweep", mode)
        B = _padded(H, rhs, cols_pad=1, fill=3.0)
        H.potrs(A, n, B, 1)
        H.sync()
        got = B.cpu().numpy()[:n, 0]
        assert np.all(np.isfinite(got))
        out.setdefault(mode, []).append(got)
    H.set_option("bwd_sweep", 1)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[1][0], out[1][1])
    if n <= 8200:
        L = np.tril(A.cpu().numpy()[:n, :n])
        ref = sla.solve_triangular(L.T, sla.solve_triangular(L, rhs[:, 0], lower=True), lower=False)
        assert np.max(np.abs(out[1][0] - ref)) / np.max(np.abs(ref)) < 1e-11


@pytest.mark.parametrize("n", [100, 128, 1024, 3333, 8192, 36000])
def test_forward_sweep_in_one_launch(H, n):
    """L y = b with one right-hand side (fvgp_hip_trsm_lower; the solve of a log-likelihood whose size leaves no padding row for
    the fused one, n % 128 == 0): the single-launch forward sweep (workgroup r owns block row r and hands y_r to the right through
    tagged granules) against the per-block launches (option fwd_sweep = 0: another order of the same sums) and scipy; the same
    bits run after run; the right-hand side is not modified beyond the result."""
    from fvgp_amd._lib import pad128
    import torch
    rng = np.random.default_rng(n + 1)
    npad = pad128(n)
    A = H.empty(npad, npad)
    A.zero_()
    g = torch.Generator(device=A.device); g.manual_seed(n + 1)
    blk = 4096
    for r0 in range(0, npad, blk):
        r1 = min(npad, r0 + blk)
        A[r0:r1, :r1] = 0.02 * torch.randn(r1 - r0, r1, dtype=torch.float64, device=A.device, generator=g) / np.sqrt(npad)
    A.diagonal().copy_(1.0 + torch.rand(npad, dtype=torch.float64, device=A.device, generator=g))
    A[n:, :] = 0.0
    A.diagonal()[n:] = 1.0
    H.invalidate_factor()
    rhs = rng.standard_normal((n, 1))
    out = {}
    try:
        for mode in (0, 1, 1):
            H.set_option("fwd_sweep", mode)
            B = _padded(H, rhs, cols_pad=1, fill=3.0)
            H.trsm_lower(A, n, B, 1)
            H.sync()
            got = B.cpu().numpy()[:n, 0]
            assert np.all(np.isfinite(got))
            out.setdefault(mode, []).append(got)
    finally:
        H.set_option("fwd_sweep", 1)
    assert np.array_equal(out[1][0], out[1][1])
    scale = np.max(np.abs(out[0][0]))
    assert np.max(np.abs(out[1][0] - out[0][0])) / scale < 1e-13
    if n <= 8192:
        L = np.tril(A.cpu().numpy()[:n, :n])
        ref = sla.solve_triangular(L, rhs[:, 0], lower=True)
        assert np.max(np.abs(out[1][0] - ref)) / np.max(np.abs(ref)) < 1e-12


def test_launch_shape_options_keep_the_result(H):
    """The options that only pick a launch shape: `tile_tables` (XCD-balanced block -> tile table vs the formula map) and
    the diagnostic stamp buffers change WHERE and WHEN a tile runs, never its arithmetic: same bits.  `lookahead_min` (look-ahead
    on at this size: two streams and a split trailing update vs wide panels on one stream), `chain_wide` = 0 (narrow panels on one stream) and  `small_tile_max(_update)` = 0 sends the chain's small products to the
    128-tile kernel, which contracts k in another order: LAPACK accuracy either way.  The diagnostic stamp buffers
    (`chain_stamps`, `leaf_stamps`) fill when set and change nothing."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for (M, N, K) in ((1024, 1024, 528), (768, 128, 1040)):
        A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
        B = torch.randn(N, K, dtype=torch.float64, device="cuda", generator=g)
        C0 = torch.randn(M, N, dtype=torch.float64, device="cuda", generator=g)
        out = {}
        try:
            for v in (1, 0):
                H.set_option("tile_tables", v)
                C = C0.clone()
                H.gemm(0, 0, 0, M, N, K, -0.75, A, B, 1.25, C)
                H.sync()
                out[v] = C
        finally:
            H.set_option("tile_tables", 1)
        assert torch.equal(out[0], out[1])
        assert float((out[1] + 0.75 * A @ B.T - 1.25 * C0).abs().max()) < 1e-11 * K
    from fvgp_amd._lib import pad128
    n = 5200
    M_ = _spd(n, 23)
    Lref = np.tril(sla.cho_factor(M_, lower=True)[0])
    npad = pad128(n)
    buf = np.zeros((npad, npad)); buf[:n, :n] = np.tril(M_)
    got = {}
    stamps = torch.zeros(8 + 4 * 4096, dtype=torch.int64, device="cuda")
    lstamps = torch.zeros(64, dtype=torch.int64, device="cuda")
    try:
        defaults = dict(lookahead_min=1 << 40, chain_wide=1, small_tile_max=160, small_tile_max_update=512, chain_stamps=0, leaf_stamps=0)
        for name, opts in (("default", {}), ("stamps", {"chain_stamps": stamps.data_ptr(), "leaf_stamps": lstamps.data_ptr()}),
                           ("lookahead", {"lookahead_min": 0}),
                           ("lookahead_stamps", {"lookahead_min": 0, "chain_stamps": stamps.data_ptr(), "leaf_stamps": lstamps.data_ptr()}),
                           ("narrow_panels", {"chain_wide": 0}),
                           ("no_small_tiles", {"small_tile_max": 0, "small_tile_max_update": 0})):
            for k, v in opts.items():
                H.set_option(k, v)
            A = H.to_device(buf)
            assert H.potrf(A, n) == 0
            got[name] = np.tril(A.cpu().numpy()[:n, :n])
            for k, v in defaults.items():
                H.set_option(k, v)
    finally:
        for k, v in dict(lookahead_min=1 << 40, chain_wide=1, small_tile_max=160, small_tile_max_update=512, chain_stamps=0, leaf_stamps=0).items():
            H.set_option(k, v)
    assert np.array_equal(got["default"], got["stamps"]) and np.array_equal(got["lookahead"], got["lookahead_stamps"])
    assert int(stamps[0]) > 0 and int(lstamps[:8].abs().sum()) > 0
    # (the default -- 4096-wide panels, each factored by one resident kernel alone on the chip -- the look-ahead schedule on two streams
    # and the narrow panels without look-ahead contract k in different orders)
    for name in ("default", "lookahead", "narrow_panels", "no_small_tiles"):
        assert np.max(np.abs(got[name] - Lref)) / np.max(np.abs(Lref)) < 1e-13


@pytest.mark.parametrize("one_stage", [1, 0])
def test_gemm_in_place_over_the_nk_operand(H, one_stage):
    """(M,K) x (N,K) with C == B, M = N = K = 128 (a public-ABI call nobody inside makes): a small-tile workgroup owns 32 rows
    of C, which are rows of B the other workgroups still read, so this aliasing must stay on the one-workgroup 128-tile."""
    rng = np.random.default_rng(77)
    a = rng.standard_normal((128, 128)); b = rng.standard_normal((128, 128))
    H.set_option("k128_kernels", one_stage)
    try:
        A = H.to_device(a); B = H.to_device(b)
        H.gemm(0, 0, 0, 128, 128, 128, 1.0, A, B, 0.0, B)
        H.sync()
    finally:
        H.set_option("k128_kernels", 1)
    ref = a @ b.T
    assert np.max(np.abs(B.cpu().numpy() - ref) / (np.abs(a) @ np.abs(b.T))) < 8 * EPS


@pytest.mark.parametrize("one_stage", [1, 0])
@pytest.mark.parametrize("lower", [0, 1])
def test_gemm_k128_chain_products(H, lower, one_stage):
    """The two K = 128 products of a panel-chain step on the small-tile kernels (one-stage fetch or eight double-buffered
    steps): the block-column update C -= A B^T, and the TRSM X = A inv(L)^T written over A."""
    rng = np.random.default_rng(20 + lower)
    H.set_option("k128_kernels", one_stage)
    try:
        M, N, K = 640, 384 if not lower else 640, 128
        a = rng.standard_normal((M, K)); b = rng.standard_normal((N, K)); C0 = rng.standard_normal((M, N))
        C = H.to_device(C0)
        H.gemm(0, 0, lower, M, N, K, -1.0, H.to_device(a), H.to_device(b), 1.0, C)
        H.sync()
        got, ref, scale = C.cpu().numpy(), C0 - a @ b.T, np.abs(a) @ np.abs(b.T) + np.abs(C0)
        for ti in range(M // 128):
            for tj in range(N // 128):
                sl = (slice(ti * 128, ti * 128 + 128), slice(tj * 128, tj * 128 + 128))
                if not lower or tj <= ti:
                    assert np.max(np.abs(got[sl] - ref[sl]) / scale[sl]) < 8 * EPS
                else:
                    assert np.array_equal(got[sl], C0[sl]), "tile above the diagonal must be untouched"
        # in place: A (M x 128 inside a wider buffer) <- A inv^T
        wide = rng.standard_normal((M, 512)); inv = np.tril(rng.standard_normal((128, 128)))
        W = H.to_device(wide)
        H.gemm(0, 0, 0, M, 128, 128, 1.0, W[:, 128:256], H.to_device(inv), 0.0, W[:, 128:256])
        H.sync()
        out = W.cpu().numpy()
        want = wide[:, 128:256] @ inv.T
        assert np.max(np.abs(out[:, 128:256] - want) / (np.abs(wide[:, 128:256]) @ np.abs(inv.T))) < 8 * EPS
        assert np.array_equal(out[:, :128], wide[:, :128]) and np.array_equal(out[:, 256:], wide[:, 256:])
    finally:
        H.set_option("k128_kernels", 1)


def test_gemm_beta_zero_and_big_k(H):
    rng = np.random.default_rng(3)
    M, N, K = 128, 128, 2048
    a = rng.standard_normal((M, K)); b = rng.standard_normal((N, K))
    C = H.to_device(np.full((M, N), np.nan))          # beta = 0 must not read C
    H.gemm(0, 0, 0, M, N, K, 1.0, H.to_device(a), H.to_device(b), 0.0, C)
    H.sync()
    got = C.cpu().numpy()
    assert np.max(np.abs(got - a @ b.T) / (np.abs(a) @ np.abs(b.T))) < 32 * EPS


KERNEL_CASES = [("rbf_ard", 3), ("matern32_ard", 2), ("matern52_ard", 3), ("rbf_ard", 1), ("matern52_ard", 5),
                ("rbf_iso", 2), ("matern32_iso", 3), ("matern52_iso", 2)]


@pytest.mark.parametrize("name,d", KERNEL_CASES)
def test_kmat_rectangular(H, name, d):
    from fvgp_amd import _lib
    rng = np.random.default_rng(5)
    x1 = rng.random((300, d)); x2 = rng.random((201, d))
    theta = np.concatenate([[1.7], 0.2 + rng.random(d if name.endswith("ard") else 1)])
    ref = orc.KERNELS[name](x1, x2, theta)
    K = H.to_device(np.full((300, 201), np.nan))
    H.kmat(_lib.KERNEL_IDS[name], H.to_device(x1), H.to_device(x2), theta, K, pad=_lib.PAD_NONE)
    H.sync()
    got = K.cpu().numpy()
    # K entries within 4 ulp sigma^2 (measured worst over 7.5e6 entries per case: RBF 1.9, Matern-3/2 3.2, Matern-5/2 3.7: profiles/r06_kmat_ulp.txt)
    assert np.max(np.abs(got - ref)) <= 4 * EPS * theta[0]
    # padded variant with zero fill
    Kp = H.to_device(np.full((384, 256), np.nan))
    H.kmat(_lib.KERNEL_IDS[name], H.to_device(x1), H.to_device(x2), theta, Kp, pad=_lib.PAD_ZERO)
    H.sync()
    gp = Kp.cpu().numpy()
    assert np.array_equal(gp[:300, :201], got)
    assert np.all(gp[300:, :] == 0) and np.all(gp[:, 201:] == 0)


@pytest.mark.parametrize("name", ["rbf_ard", "matern32_ard", "matern52_ard"])
def test_kmat_nan_and_far_points_follow_numpy(H, name):
    """What numpy's exp / sqrt make of unusual inputs (kernels.py:16-33,98-118,166-188,461-481), the device assembly makes too:
    a NaN coordinate gives a NaN row and column (not a finite covariance), points so far apart that the squared scaled distance
    overflows give exactly 0 for the RBF (exp(-inf)) and numpy's own inf * 0 = NaN for the Matern forms, large finite distances
    give exactly 0."""
    from fvgp_amd import _lib
    rng = np.random.default_rng(15)
    x1 = rng.random((260, 3)); x2 = rng.random((140, 3))
    x1[7, 1] = np.nan                      # a NaN point
    x1[9] = 1e160                          # squared distance overflows to inf
    x2[3, 0] = -1e100                      # huge but finite distance
    x2[5, 2] = np.nan
    theta = np.array([1.3, 0.3, 0.4, 0.5])
    with np.errstate(all="ignore"):
        ref = orc.KERNELS[name](x1, x2, theta)
    K = H.to_device(np.full((260, 140), 7.0))
    H.kmat(_lib.KERNEL_IDS[name], H.to_device(x1), H.to_device(x2), theta, K, pad=_lib.PAD_NONE)
    H.sync()
    got = K.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.all(np.isnan(got[7])) and np.all(np.isnan(got[:, 5]))
    ok = ~np.isnan(ref)
    assert np.max(np.abs(got[ok] - ref[ok])) <= 4 * EPS * theta[0]
    far = np.zeros_like(ok); far[:, 3] = True; far &= ok
    assert np.all(got[far] == 0.0) and np.all(ref[far] == 0.0)


@pytest.mark.parametrize("name,d", KERNEL_CASES[:4])
def test_kmat_lower_padded_with_noise(H, name, d):
    from fvgp_amd import _lib
    rng = np.random.default_rng(6)
    n = 333
    x = rng.random((n, d))
    theta = np.concatenate([[0.9], 0.25 + rng.random(d)])
    v = 0.01 + rng.random(n)
    ref = orc.addKV(orc.KERNELS[name](x, x, theta), v)
    K = H.to_device(np.full((384, 384), np.nan))
    xd = H.to_device(x)
    H.kmat(_lib.KERNEL_IDS[name], xd, xd, theta, K, vdiag=H.to_device(v), uplo=_lib.LOWER, pad=_lib.PAD_IDENTITY)
    H.sync()
    got = K.cpu().numpy()
    il = np.tril_indices(n)
    assert np.max(np.abs(got[:n, :n][il] - ref[il])) <= 4 * EPS * theta[0]
    # padding = identity (lower tiles), and tiles strictly above the block diagonal untouched (NaN)
    assert np.array_equal(np.tril(got[n:, :])[:, :n], np.zeros((384 - n, n)))
    assert np.array_equal(np.diag(got)[n:], np.ones(384 - n))
    assert np.all(np.isnan(got[:128, 128:]))


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n))
    return B @ B.T + n * np.eye(n)


@pytest.mark.parametrize("n", [96, 128, 200, 512, 1000, 1537])
@pytest.mark.parametrize("outer", [128, 512])
def test_potrf_solve_logdet(H, n, outer):
    """calculate_Chol_factor / _solve / _logdet (gp_lin_alg.py:237-360) on B B^T + n I."""
    from fvgp_amd._lib import pad128
    M = _spd(n, n)
    H.set_option("outer_block", outer)
    npad = pad128(n)
    buf = np.full((npad, npad), np.nan)       # strict upper left as NaN: the kernels must never read it
    buf[:n, :n] = np.tril(M) + np.triu(np.full((n, n), np.nan), 1)
    buf[n:, :] = 0.0
    buf[n:, n:] = np.tril(np.eye(npad - n)) + np.triu(np.full((npad - n, npad - n), np.nan), 1)
    A = H.to_device(buf)
    info = H.potrf(A, n)
    assert info == 0
    L = np.tril(A.cpu().numpy()[:n, :n])
    Lref = np.tril(sla.cho_factor(M, lower=True)[0])
    assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-13
    np.testing.assert_allclose(H.logdet(A, n), 2 * np.sum(np.log(np.diag(Lref))), rtol=1e-13)
    rng = np.random.default_rng(n + 1)
    for nrhs in (1, 3, 8):
        rhs = rng.standard_normal((n, nrhs))
        B = _padded(H, rhs, cols_pad=nrhs, fill=7.0)   # junk in padding rows must be cleared
        H.potrs(A, n, B, nrhs)
        H.sync()
        ref = sla.cho_solve((Lref, True), rhs)
        got = B.cpu().numpy()
        assert np.max(np.abs(got[:n] - ref)) / np.max(np.abs(ref)) < 1e-12
        assert np.all(got[n:] == 0)
    rhs = rng.standard_normal((n, 128))
    B = _padded(H, rhs)
    H.potrs(A, n, B, 128)
    H.sync()
    ref = sla.cho_solve((Lref, True), rhs)
    assert np.max(np.abs(B.cpu().numpy()[:n] - ref)) / np.max(np.abs(ref)) < 1e-12
    B = _padded(H, rhs)
    H.trsm_lower(A, n, B, 128)
    H.sync()
    ref = sla.solve_triangular(Lref, rhs, lower=True)
    assert np.max(np.abs(B.cpu().numpy()[:n] - ref)) / np.max(np.abs(ref)) < 1e-12
    H.set_option("outer_block", 1024)          # the library default (common.h)


@pytest.mark.parametrize("n,nrhs", [(2500, 128), (4200, 384), (2048, 1024)])
def test_trsm_lower_few_columns_against_a_long_factor(H, n, nrhs):
    """fvgp_hip_trsm_lower with 128 .. 1024 columns and at least 2048 rows (the new rows of an append, gp_lin_alg.py:1310-1477):
    the right-hand sides are transposed and swept with the inverted diagonal blocks; against scipy and against the per-128-block
    substitution (option block_inverses = 0)."""
    from fvgp_amd._lib import pad128
    M = _spd(n, n + 3)
    npad = pad128(n)
    buf = np.zeros((npad, npad))
    buf[:n, :n] = np.tril(M)
    buf[n:, n:] = np.eye(npad - n)
    A = H.to_device(buf)
    assert H.potrf(A, n) == 0
    Lref = np.tril(sla.cho_factor(M, lower=True)[0])
    rhs = np.random.default_rng(n).standard_normal((n, nrhs))
    ref = sla.solve_triangular(Lref, rhs, lower=True)
    got = {}
    try:
        for mode in (1, 0):
            H.set_option("block_inverses", mode)
            B = _padded(H, rhs, fill=5.0)
            H.trsm_lower(A, n, B, nrhs)
            H.sync()
            got[mode] = B.cpu().numpy()
            assert np.all(got[mode][n:] == 0)
    finally:
        H.set_option("block_inverses", 1)
    for mode in (1, 0):
        assert np.max(np.abs(got[mode][:n] - ref)) / np.max(np.abs(ref)) < 1e-12
    assert np.max(np.abs(got[1] - got[0])) / np.max(np.abs(ref)) < 1e-12


@pytest.mark.parametrize("sched", [dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=1024, big_threshold=0, inner_block=256, lookahead=1),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=256, outer_block_big=1024, big_threshold=1500, inner_block=128, lookahead=1),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=256, outer_block_big=768, big_threshold=0, inner_block=512, lookahead=1),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=1536, big_threshold=1000, inner_block=512, lookahead=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=1024, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=1024, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1, leaf_tiles=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1, leaf_tiles_rows=1024),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=256, outer_block_big=768, big_threshold=0, inner_block=256, lookahead=1, panel_recursive=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=1536, big_threshold=1000, inner_block=512, lookahead=0, panel_recursive=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1, panel_chain=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=512, outer_block_big=2048, big_threshold=1000, inner_block=512, lookahead=1, panel_chain=1, panel_chain_min=0),
                                   dict(chain_wide=0, lookahead_min=4608, outer_block=256, outer_block_big=1024, big_threshold=24576, inner_block=512, lookahead=0, panel_chain=1, panel_chain_min=0, outer_block_small=128, small_threshold=1024),
                                   dict(chain_wide=0, outer_block=256, outer_block_big=768, big_threshold=0, inner_block=512, lookahead=1, lookahead_min=0, panel_chain=1, panel_chain_min=0, cols_split=1, cols_split_rows=100000),
                                   dict(chain_wide=0, outer_block=512, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1, lookahead_min=0, panel_chain=1, panel_chain_min=0, cols_split=0),
                                   dict(),
                                   dict(wide_block=1024, wide_block_big=1024),
                                   dict(wide_block=512, wide_block_big=2048, wide_threshold=1500),
                                   dict(wide_block=2048, wide_block_big=2048, wide_inner=512, wide_inner_rows=0),
                                   dict(wide_block=4096, wide_block_big=4096, wide_inner=1024, wide_inner_rows=2048),
                                   dict(schedule=0), dict(schedule=1), dict(schedule=2), dict(schedule=0, chain_ahead=3)])
def test_potrf_panel_schedules_agree(H, sched):
    """Every panel schedule (regular / wide panels, sub-panels of a third block size, with and without look-ahead; the
    chain's TRSM by the inverted 128-block, by substitution with the 16 x 16 tile inverses, or switching between the two
    on the way down; panels by recursive halving (the default) or by 128-column steps inside sub-panels; the chain as three launches
    per 128 columns or as ONE resident kernel per panel, chain.hip, for panels of 128 ... 2048 columns, its rows below the square
    updated before it starts or beside it behind a flag in memory; the default: panels of up to 4096 columns, each ONE resident
    kernel alone on the chip with a workgroup per block, optionally in sub-panels with the trailing update's kernel in between)
    is the same factorisation: compare with LAPACK on one matrix."""
    from fvgp_amd._lib import pad128
    n = 3000
    M = _spd(n, 17)
    Lref = np.tril(sla.cho_factor(M, lower=True)[0])
    for k, v in sched.items():
        H.set_option(k, v)
    try:
        npad = pad128(n)
        buf = np.zeros((npad, npad))
        buf[:n, :n] = np.tril(M)
        A = H.to_device(buf)
        assert H.potrf(A, n) == 0
        L = np.tril(A.cpu().numpy()[:n, :n])
        assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-13
        np.testing.assert_allclose(H.logdet(A, n), 2 * np.sum(np.log(np.diag(Lref))), rtol=1e-13)
        # the 128-block inverses a solve takes after the factorisation belong to this factor whichever way the chain went
        b = np.random.default_rng(3).standard_normal((npad, 1)); b[n:] = 0.0
        B = H.to_device(b)
        H.potrs(A, n, B, 1)
        H.sync()
        want = sla.cho_solve((Lref, True), b[:n, 0])
        assert np.max(np.abs(B.cpu().numpy()[:n, 0] - want)) / np.max(np.abs(want)) < 1e-9
    finally:
        _restore_schedule_defaults(H)


def _restore_schedule_defaults(H):
    """the library's own defaults (fvgp_amd/csrc/common.h) for every key the schedule tests touch: the handle is shared by the module"""
    for k, v in dict(outer_block=1024, outer_block_big=2048, big_threshold=24576, inner_block=512, lookahead=1, leaf_tiles=1,
                     leaf_tiles_rows=4096, panel_recursive=1, panel_chain=1, panel_chain_min=4096, outer_block_small=512,
                     small_threshold=12288, lookahead_min=1 << 40, cols_split=1, cols_split_rows=8192, chain_wide=1, wide_block=4096,
                     wide_block_big=4096, wide_threshold=1 << 30, wide_inner=2048, wide_inner_rows=16384, chain_ahead=0).items():
        H.set_option(k, v)
    H.set_option("schedule", 0)


def test_panels_wider_than_the_resident_kernel_takes(H):
    """outer_block accepts any multiple of 128; the resident panel kernel has flag words for 32 block columns (4096): a wider
    panel -- here ONE panel of 8192 columns over a 9000-row matrix, with enough rows below it for the resident kernel to be
    chosen -- is factored by the launch-per-step chain instead of failing the factorisation, and matches LAPACK."""
    from fvgp_amd._lib import pad128
    n = 9000
    M = _spd(n, 23)
    Lref = np.tril(sla.cho_factor(M, lower=True)[0])
    try:
        for k, v in dict(chain_wide=0, lookahead_min=4608, outer_block=8192, outer_block_big=0, outer_block_small=0).items():
            H.set_option(k, v)
        npad = pad128(n)
        buf = np.zeros((npad, npad))
        buf[:n, :n] = np.tril(M)
        A = H.to_device(buf)
        assert H.potrf(A, n) == 0
        L = np.tril(A.cpu().numpy()[:n, :n])
        assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-13
    finally:
        _restore_schedule_defaults(H)


def test_potrf_nonpd_info(H):
    """dpotrf info: order of the first non-positive leading minor (tests/test_fvgp.py:4653-4665)."""
    fx = load_golden("G7_nonpd.npz")
    A = _padded(H, np.tril(fx["M"]))
    A[96:, 96:] = H.to_device(np.eye(32))
    assert H.potrf(A, 96) == int(fx["info"])
    # and a failure in a later diagonal block of a larger matrix
    M = _spd(700, 9)
    M[444, 444] = -3.0
    A = _padded(H, np.tril(M))
    A[700:, 700:] = H.to_device(np.eye(68))
    try:
        sla.cho_factor(M, lower=True)
        raise AssertionError("scipy accepted it?")
    except np.linalg.LinAlgError as e:
        assert "445-th leading minor" in str(e)
    assert H.potrf(A, 700) == 445


def test_golden_lin_alg_block(H):
    fx = load_golden("G7_nonpd.npz")
    A = _padded(H, np.tril(fx["Mok"]))
    A[96:, 96:] = H.to_device(np.eye(32))
    assert H.potrf(A, 96) == 0
    np.testing.assert_allclose(np.tril(A.cpu().numpy()[:96, :96]), fx["Lok"], rtol=0, atol=1e-13 * np.max(fx["Lok"]))
    B = _padded(H, fx["rhs"], cols_pad=3)
    H.potrs(A, 96, B, 3)
    H.sync()
    np.testing.assert_allclose(B.cpu().numpy()[:96], fx["sol"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(H.logdet(A, 96), float(fx["logdet"]), rtol=1e-13)


@pytest.mark.parametrize("n", [128, 300, 900, 2500, 3333])
def test_potri(H, n):
    from fvgp_amd._lib import pad128
    M = _spd(n, 40 + n)
    npad = pad128(n)
    buf = np.zeros((npad, npad)); buf[:n, :n] = np.tril(M); buf[n:, n:] = np.eye(npad - n)
    A = H.to_device(buf)
    assert H.potrf(A, n) == 0
    W = H.empty(npad, npad)
    H.potri(A, n, W)
    H.sync()
    inv = np.linalg.inv(M)
    got = np.tril(A.cpu().numpy()[:n, :n])
    assert np.max(np.abs(got - np.tril(inv))) / np.max(np.abs(inv)) < 1e-11
    H.symmetrize(A, n)
    H.sync()
    full = A.cpu().numpy()[:n, :n]
    assert np.max(np.abs(full - inv)) / np.max(np.abs(inv)) < 1e-11
    # the round-2 schedule ((K,N) / (K,M) operands) gives the same inverse
    A2 = H.to_device(buf)
    assert H.potrf(A2, n) == 0
    H.set_option("potri_kminor", 0)
    try:
        H.potri(A2, n, W)
    finally:
        H.set_option("potri_kminor", 1)
    H.sync()
    got2 = np.tril(A2.cpu().numpy()[:n, :n])
    assert np.max(np.abs(got2 - got)) / np.max(np.abs(inv)) < 1e-12


GOLD = ["G1_rbf_n500_d1.npz", "G2_rbf_n512_d3.npz", "G3_matern52_n512_d3.npz", "G6_rbf_2col_n300_d3.npz",
        "G5_fvgp_4x64.npz", "G5n_fvgp_4x64_nan.npz"]


@pytest.mark.parametrize("name", GOLD)
def test_fused_loglik_grad_against_reference_vectors(H, name):
    """fvgp_hip_loglik / _loglik_grad against the values the reference itself produced."""
    from fvgp_amd import _lib
    fx = load_golden(name)
    x = fx["x"]; y = fx["y"].reshape(len(x), -1); n = len(x)
    kid = _lib.KERNEL_IDS[str(fx["kernel"])]
    ym = y - np.mean(y)
    npad = _lib.pad128(n)
    xd, vd, ymd = H.to_device(x), H.to_device(fx["noise_variances"]), H.to_device(ym)
    KV = H.empty(npad, npad); W = H.empty(npad, npad); alpha = H.empty(npad, y.shape[1])
    for theta, ll_ref in [(fx["theta"], float(fx["loglik_theta"]))] + list(zip(fx["thetas"], fx["logliks"])):
        ll, logdet, quad, info = H.loglik(kid, xd, theta, vd, ymd, KV, alpha)
        assert info == 0
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)       # log-lik rel <= 1e-10
    ll, logdet, quad, info = H.loglik(kid, xd, fx["theta"], vd, ymd, KV, alpha)
    np.testing.assert_allclose(logdet, float(fx["logdet"]), rtol=1e-10)
    a = alpha.cpu().numpy()[:n]
    assert np.max(np.abs(a - fx["KVinvY"])) / np.max(np.abs(fx["KVinvY"])) < 1e-8   # KVinvY rel <= 1e-8
    Lg = np.tril(KV.cpu().numpy()[:n, :n])
    np.testing.assert_allclose(np.diag(Lg), fx["L_diag"], rtol=1e-10)
    np.testing.assert_allclose(Lg[-1, :], fx["L_row_last"], rtol=0, atol=1e-10 * np.max(np.abs(fx["L_diag"])))
    g = H.loglik_grad(kid, xd, fx["theta"], alpha, y.shape[1], 0, KV, W)
    np.testing.assert_allclose(g, fx["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad"])))
    if "grad_c1" in fx:
        ll, logdet, quad, info = H.loglik(kid, xd, fx["theta"], vd, ymd, KV, alpha)
        g = H.loglik_grad(kid, xd, fx["theta"], alpha, y.shape[1], 1, KV, W)
        np.testing.assert_allclose(g, fx["grad_c1"], rtol=1e-8, atol=1e-9 * np.max(np.abs(fx["grad_c1"])))


@pytest.mark.parametrize("n,ncol", [(1024, 1), (2048, 2), (4089, 8), (640, 1), (4096, 1)])
def test_fused_forward_solve_for_sizes_without_padding_rows(H, n, ncol):
    """n a multiple of 128 (or leaving fewer padding rows than y has columns): with a square scratch of fvgp_hip_loglik_dim(n, ncol)
    the appended (y-m)^T rows take a block row of their own and the forward solve rides in the factorisation as for every other n
    (gp_kv.py:589-593); with a scratch of only padded_dim(n) the forward solve is a sweep of its own.  Both against the oracle, the
    factor handed back clean (identity on every padding row), the backward solve and a posterior on it."""
    from fvgp_amd import _lib
    rng = np.random.default_rng(n + ncol)
    x = rng.random((n, 3))
    y = np.stack([np.sin((3.0 + c) * x.sum(axis=1)) for c in range(ncol)], axis=1) + 0.1 * rng.standard_normal((n, ncol))
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.35, 0.25])
    ref = orc.OracleGP(x, y, theta, nv, kernel="rbf_ard")
    ym = y - np.mean(y)
    npad, nfull = _lib.pad128(n), _lib.loglik_dim(n, ncol)
    assert nfull == npad + 128 and _lib.lib().fvgp_hip_loglik_dim(n, ncol) == nfull
    xd, vd, ymd = H.to_device(x), H.to_device(nv), H.to_device(ym)
    outs = []
    for dim in (nfull, npad):
        KV = H.to_device(np.full((dim, dim), np.nan)); alpha = H.empty(npad, ncol)
        ll, logdet, quad, info = H.loglik(0, xd, theta, vd, ymd, KV, alpha)
        H.sync()
        assert info == 0
        np.testing.assert_allclose(ll, ref.log_likelihood(), rtol=1e-10)
        np.testing.assert_allclose(logdet, ref.logdet_KV, rtol=1e-10)
        a = alpha.cpu().numpy()[:n]
        assert np.max(np.abs(a - ref.KVinvY)) <= 1e-8 * np.max(np.abs(ref.KVinvY))
        Lg = KV.cpu().numpy()
        np.testing.assert_allclose(np.diag(Lg)[:n], np.diag(ref.Chol_factor), rtol=1e-10)
        if dim > n:                                                  # identity on the padding: lower part of every padding row
            assert np.array_equal(np.tril(Lg[n:, :])[:, :n], np.zeros((dim - n, n)))
            assert np.array_equal(np.tril(Lg[n:, n:]), np.eye(dim - n))
        # likelihood only (no alpha): allowed where the forward solve is fused
        if dim == nfull:
            ll2 = H.loglik(0, xd, theta, vd, ymd, KV, None)[0]
            assert ll2 == ll
        # the factor serves a solve and a posterior afterwards
        xp = rng.random((5, 3)); Pp = _lib.pad128(5)
        kx = H.empty(npad, Pp); mean = H.empty(5, ncol); var = H.empty(5)
        H.posterior(0, xd, theta, KV, alpha, ncol, H.to_device(xp), kx, mean_out=mean, var_out=var)
        H.sync()
        np.testing.assert_allclose(mean.cpu().numpy() + np.mean(y), np.asarray(ref.posterior_mean(xp)["m(x)"]).reshape(5, ncol), rtol=1e-8, atol=1e-10)
        outs.append((ll, logdet))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-12 * abs(outs[0][0])


@pytest.mark.parametrize("name", GOLD[:4])
def test_posterior_against_reference_vectors(H, name):
    from fvgp_amd import _lib
    fx = load_golden(name)
    x = fx["x"]; y = fx["y"].reshape(len(x), -1); n = len(x); c = y.shape[1]
    kid = _lib.KERNEL_IDS[str(fx["kernel"])]
    m = np.mean(y)
    npad = _lib.pad128(n)
    xd, vd, ymd = H.to_device(x), H.to_device(fx["noise_variances"]), H.to_device(y - m)
    KV = H.empty(npad, npad); alpha = H.empty(npad, c)
    H.loglik(kid, xd, fx["theta"], vd, ymd, KV, alpha)
    xp = fx["x_pred"]; P = len(xp); Pp = _lib.pad128(P)
    kx = H.empty(npad, Pp); mean = H.empty(P, c); var = H.empty(P); S = H.empty(Pp, Pp)
    H.posterior(kid, xd, fx["theta"], KV, alpha, c, H.to_device(xp), kx, mean, var, S)
    H.sync()
    pm = mean.cpu().numpy() + m
    np.testing.assert_allclose(np.squeeze(pm), fx["pm_flat"], rtol=1e-8, atol=1e-10)
    Sg = S.cpu().numpy()[:P, :P]
    assert np.max(np.abs(Sg - fx["pS_flat"])) < 1e-10 * fx["theta"][0] + 1e-12
    vflat = fx["pv_flat"] if fx["pv_flat"].ndim == 1 else fx["pv_flat"][:, 0]
    assert np.max(np.abs(var.cpu().numpy() - vflat)) < 1e-10 * fx["theta"][0] + 1e-12   # variance abs <= 1e-10 sigma^2


def test_loglik_matches_oracle_medium(H):
    """N = 3000, d = 3 RBF on the bench generator: HIP path vs the oracle on the same inputs."""
    from fvgp_amd import _lib
    n = 3000
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    ref, _ = orc.log_likelihood_once(x, y, nv, theta, "rbf_ard")
    npad = _lib.pad128(n)
    ym = (y - np.mean(y)).reshape(n, 1)
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    ll, logdet, quad, info = H.loglik(0, H.to_device(x), theta, H.to_device(nv), H.to_device(ym), KV, alpha)
    assert info == 0
    np.testing.assert_allclose(ll, ref, rtol=1e-10)


def test_scheduling_mechanisms_do_not_change_a_bit(H):
    """N = 9000 (look-ahead on, the chain runs under the trailing update): the cooperative yield of the update's waves to the
    leaf / the chain's K = 128 kernels, and the single-launch backward sweep, only change WHEN things run -- log-likelihood,
    log-determinant, quadratic form and alpha must come out bit-identical with each of them off.  The first and the last
    configuration are the same: the resident panel kernel's in-launch hand-offs (chain.hip) give the same bits run after run;
    with the panel kernel off (the three-launch chain: another order of the same sums) the results agree to rounding."""
    from fvgp_amd import _lib
    n = 9000
    x, y = synth(n, 3)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    npad = _lib.pad128(n)
    xd = H.to_device(x); vd = H.to_device(np.full(n, 0.01))
    ymd = H.zeros(npad, 1); ymd[:n, 0] = H.to_device(y - np.mean(y))
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    out = []
    try:
        for ly, cy, bs in ((1, 1, 1), (0, 0, 1), (1, 0, 0), (1, 1, 1)):
            H.set_option("leaf_yield", ly); H.set_option("chain_yield", cy); H.set_option("bwd_sweep", bs)
            ll, logdet, quad, info = H.loglik(0, xd, theta, vd, ymd, KV, alpha)
            H.sync()
            assert info == 0
            out.append((ll, logdet, quad, alpha[:n, 0].cpu().numpy().copy()))
    finally:
        H.set_option("leaf_yield", 1); H.set_option("chain_yield", 1); H.set_option("bwd_sweep", 1)
    for o in out[1:]:
        assert o[:3] == out[0][:3]
        assert np.array_equal(o[3], out[0][3])
    try:
        H.set_option("panel_chain", 0)
        ll, logdet, quad, info = H.loglik(0, xd, theta, vd, ymd, KV, alpha)
        H.sync()
    finally:
        H.set_option("panel_chain", 1)
    assert info == 0
    assert abs(ll - out[0][0]) <= 1e-12 * abs(ll) and abs(logdet - out[0][1]) <= 1e-12 * abs(logdet)
    assert np.max(np.abs(alpha[:n, 0].cpu().numpy() - out[0][3])) <= 1e-9 * np.max(np.abs(out[0][3]))


@pytest.mark.parametrize("n", [8000, 12000])
def test_resident_panel_kernel_soak(H, n):
    """30 evaluations of the same theta (the resident panel kernel of chain.hip, a workgroup per block, hundreds of in-launch hand-offs
    per panel; at n = 12000 under the look-ahead schedule instead: the kernel beside a full trailing update, uneven load, consumers
    with warm caches), the matrix buffer poisoned with NaN before each: log-likelihood, log-det, quadratic
    form, alpha and every entry of L must come out with the same bits each time -- a stale or torn in-launch hand-off would change
    some of them -- and agree with LAPACK's answer (the oracle) to rounding."""
    import torch
    from fvgp_amd import _lib
    from oracle import fvgp_oracle as orc
    x, y = synth(n, 3)
    theta = np.array([1.0, 0.3, 0.3, 0.3])
    npad = _lib.pad128(n)
    xd = H.to_device(x); vd = H.to_device(np.full(n, 0.01))
    ymd = H.zeros(npad, 1); ymd[:n, 0] = H.to_device(y - np.mean(y))
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    ref = None
    differ = 0
    if n == 12000:
        H.set_option("lookahead_min", 4608)
    for _ in range(30):
        KV.fill_(float("nan"))
        out = H.loglik(0, xd, theta, vd, ymd, KV, alpha)
        H.sync()
        assert out[3] == 0
        a, L = alpha[:n, 0].clone(), KV[:n, :n].tril()
        if ref is None:
            ref = (out, a, L)
        elif not (out == ref[0] and torch.equal(a, ref[1]) and torch.equal(L, ref[2])):
            differ += 1
        del a, L
    assert differ == 0
    # the same evaluation with a checksum on every hand-off of the resident kernel (option "chain_verify": the producer sums the bit
    # patterns of what it stores, every consumer sums what ARRIVED in its LDS images / registers): no mismatch in thousands of
    # comparisons, and the same bits as the plain runs
    try:
        H.set_option("chain_verify", 1)
        H.chain_verify_counts()
        vdiffer = []
        for t in range(10):
            KV.fill_(float("nan"))
            out = H.loglik(0, xd, theta, vd, ymd, KV, alpha)
            H.sync()
            if not (out == ref[0] and torch.equal(alpha[:n, 0], ref[1]) and torch.equal(KV[:n, :n].tril(), ref[2])):
                vdiffer.append((t, out, int((KV[:n, :n].tril() != ref[2]).sum().item())))
        bad, checks = H.chain_verify_counts()
    finally:
        H.set_option("chain_verify", 0)
        H.set_option("lookahead_min", 1 << 40)
    assert bad == 0 and checks > 1000 and not vdiffer, (bad, checks, vdiffer, ref[0])
    want, _ = orc.log_likelihood_once(x, y, np.full(n, 0.01), theta, "rbf_ard")
    np.testing.assert_allclose(ref[0][0], want, rtol=1e-10)


def test_serialised_kernels_do_not_hang_the_split_update(tmp_path):
    """With kernels serialised (AMD_SERIALIZE_KERNEL=3; hardware-counter profilers do the same) a kernel that waits in memory for a
    kernel of another stream would only end by its timeout: the handle probes once whether its two streams run concurrently and
    keeps the split column update off when they do not -- the evaluation completes and agrees with the concurrent one."""
    import subprocess
    import sys
    code = f"""
import sys, json
sys.path.insert(0, {ROOT!r})
import numpy as np
from fvgp_amd import _lib
n = 9000
rng = np.random.default_rng(20240501); x = rng.random((n, 3)); y = np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
H = _lib.Handle(0)
npad = _lib.pad128(n)
ymd = H.zeros(npad, 1); ymd[:n, 0] = H.to_device(y - np.mean(y))
KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
out = H.loglik(0, H.to_device(x), np.array([1.0, 0.3, 0.3, 0.3]), H.to_device(np.full(n, 0.01)), ymd, KV, alpha)
print(json.dumps(list(out)))
"""
    outs = []
    for env_extra in ({}, {"AMD_SERIALIZE_KERNEL": "3"}):
        env = dict(os.environ, **env_extra)
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert outs[0][3] == 0 and outs[1][3] == 0
    assert abs(outs[0][0] - outs[1][0]) <= 1e-12 * abs(outs[0][0])


# ---- row-sharded building blocks (fvgp_amd/dist.py) ------------------------------------------------

@pytest.mark.parametrize("ranks,off,b_off", [(1, 0, 0), (3, 2, 4), (2, -3, 1)])
def test_syrk_rowshard_predicate_and_gathered_operand(H, ranks, off, b_off):
    """C[ti][tj] -= A[ti] B[blk(tj)]^T exactly on the tiles tj <= ti*ranks + off, with B read in the order an
    all-gather leaves it (chunk q = cyclic blocks q, q+ranks, ..)."""
    rng = np.random.default_rng(5)
    tm, tn, K = 3, 7, 256
    b_blocks = -(-(b_off + tn) // ranks)
    A = rng.standard_normal((tm * 128, K))
    Bc = rng.standard_normal(((b_off + tn + ranks) * 128, K))          # blocks in cyclic (global) order
    Bg = np.zeros((ranks * b_blocks * 128, K))                            # as gathered
    for idx in range(ranks * b_blocks):
        if (idx + 1) * 128 <= Bc.shape[0]:
            q, j = idx % ranks, idx // ranks
            Bg[(q * b_blocks + j) * 128:(q * b_blocks + j + 1) * 128] = Bc[idx * 128:(idx + 1) * 128]
    C0 = rng.standard_normal((tm * 128, tn * 128))
    Cd = H.to_device(C0)
    H.syrk_rowshard(tm * 128, tn * 128, K, H.to_device(A), H.to_device(Bg), Cd, ranks, off, ranks, b_blocks if ranks > 1 else 0, b_off)
    H.sync()
    want = C0.copy()
    for ti in range(tm):
        for tj in range(tn):
            if tj <= ti * ranks + off:
                blk = Bc[(tj + b_off) * 128:(tj + b_off + 1) * 128]
                want[ti * 128:(ti + 1) * 128, tj * 128:(tj + 1) * 128] -= A[ti * 128:(ti + 1) * 128] @ blk.T
    got = Cd.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)
    untouched = np.ones_like(C0, dtype=bool)
    for ti in range(tm):
        for tj in range(tn):
            if tj <= ti * ranks + off:
                untouched[ti * 128:(ti + 1) * 128, tj * 128:(tj + 1) * 128] = False
    assert np.array_equal(got[untouched], C0[untouched])                  # tiles outside the predicate: bit-identical


def test_syrk_rowshard_rejects_columns_past_the_gather(H):
    from fvgp_amd._lib import HipExtensionError
    Z = H.zeros(128, 512)
    with pytest.raises(HipExtensionError):
        H.syrk_rowshard(128, 512, 128, H.zeros(128, 128), H.zeros(2 * 128, 128), Z, 2, 0, 2, 1, 0)   # 4 tile cols, 2 blocks


def test_potrf_dev_and_panel_trsm(H):
    """enqueue-only potrf (info / log-det stay on the device) and the panel solve X = P L^-T."""
    import torch
    rng = np.random.default_rng(9)
    n = 384
    G = rng.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    Dd = H.to_device(S)
    info = torch.full((1,), 7, dtype=torch.int32, device="cuda:0")
    ld = H.zeros(1)
    H.potrf_dev(Dd, n, n - 5, info, ld)
    Pm = rng.standard_normal((256, n))
    Pd = H.to_device(Pm)
    H.panel_trsm(Dd, n, Pd, 256)
    H.sync()
    L = np.linalg.cholesky(S)
    assert int(info.item()) == 0
    np.testing.assert_allclose(np.tril(Dd.cpu().numpy()), L, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(float(ld.item()), 2 * np.log(np.diag(L)[:n - 5]).sum(), rtol=1e-12)
    np.testing.assert_allclose(Pd.cpu().numpy(), sla.solve_triangular(L, Pm.T, lower=True).T, rtol=1e-10, atol=1e-12)
    # not positive definite: info = order of the failing minor, on the device
    S2 = S.copy(); S2[200, 200] = -1.0
    D2 = H.to_device(S2)
    H.potrf_dev(D2, n, n, info, ld)
    H.sync()
    assert int(info.item()) == 201


@pytest.mark.parametrize("w,extra,n_valid", [(384, 256, 384), (128, 0, 100), (1024, 640, 1024)])
def test_panel_potrf_dev_tall_panel(H, w, extra, n_valid):
    """diagonal block on top, rows below solved against it; info and log-det stay on the device."""
    import torch
    rng = np.random.default_rng(11)
    G = rng.standard_normal((w, w))
    S = G @ G.T + w * np.eye(w)
    S[n_valid:, :] = 0.0; S[:, n_valid:] = 0.0
    S[np.arange(n_valid, w), np.arange(n_valid, w)] = 1.0            # identity padding below n_valid
    Pm = rng.standard_normal((extra, w))
    T = H.to_device(np.vstack([S, Pm]))
    info = torch.full((1,), 9, dtype=torch.int32, device="cuda:0")
    ld = H.zeros(1)
    H.panel_potrf_dev(T, w, w + extra, n_valid, info, ld)
    H.sync()
    L = np.linalg.cholesky(S)
    got = T.cpu().numpy()
    assert int(info.item()) == 0
    np.testing.assert_allclose(np.tril(got[:w]), L, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(float(ld.item()), 2 * np.log(np.diag(L)[:n_valid]).sum(), rtol=1e-12)
    if extra:
        np.testing.assert_allclose(got[w:], sla.solve_triangular(L, Pm.T, lower=True).T, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("n,P", [(2700, 300), (1024, 130), (3200, 5), (1300, 8200), (2700, 900), (3200, 700)])
def test_posterior_block_inverse_substitution(H, n, P):
    """The many-point posterior substitutes with the inverted 2048 x 2048 (more than 1024 points: 1024 x 1024) diagonal blocks
    (doubling from the 128-block inverses; n = 2700 leaves a last block of 5 x 128, n = 1024 a single one; P = 8200 has enough output tiles per
    block that nothing is split over K; P = 900 / 700 pad to 1024 / 768 rows, which run as two halves on two streams):
    against the oracle's cho_solve
    (gp_posterior.py:120-136,229-288) and against the 128-step substitution (option block_inverses = 0)."""
    from fvgp_amd import _lib
    x, y = synth(n, 3)
    nv = np.full(n, 0.01)
    theta = np.array([1.3, 0.35, 0.3, 0.4])
    ref = orc.OracleGP(x, y, theta, noise_variances=nv, kernel="rbf_ard")
    xp = np.random.default_rng(5).random((P, 3))
    want_m = ref.posterior_mean(xp)["m(x)"]
    want_S = ref.posterior_covariance(xp)["S"]
    npad, Pp = _lib.pad128(n), _lib.pad128(P)
    xd, vd = H.to_device(x), H.to_device(nv)
    ymd = H.to_device((y - np.mean(y)).reshape(n, 1))
    KV = H.empty(npad, npad); alpha = H.empty(npad, 1)
    H.loglik(0, xd, theta, vd, ymd, KV, alpha)
    got = {}
    for mode in (1, 0):
        H.set_option("block_inverses", mode)
        kx = H.empty(npad, Pp); mean = H.empty(P, 1); var = H.empty(P); S = H.empty(Pp, Pp)
        H.posterior(0, xd, theta, KV, alpha, 1, H.to_device(xp), kx, mean, var, S)
        H.sync()
        got[mode] = (mean.cpu().numpy()[:, 0] + np.mean(y), S.cpu().numpy()[:P, :P], var.cpu().numpy())
    H.set_option("block_inverses", 1)
    if 512 <= Pp <= 1024:            # the two halves of the points on two streams (the default at these sizes) against one stream
        try:
            H.set_option("posterior_halves", 0)
            kx = H.empty(npad, Pp); mean = H.empty(P, 1); var = H.empty(P); S = H.empty(Pp, Pp)
            H.posterior(0, xd, theta, KV, alpha, 1, H.to_device(xp), kx, mean, var, S)
            H.sync()
        finally:
            H.set_option("posterior_halves", 1)
        assert np.max(np.abs(S.cpu().numpy()[:P, :P] - got[1][1])) < 1e-11 * theta[0]
    if Pp <= 1024:                   # 2048-wide inverted blocks (the default up to 1024 points) against 1024-wide ones
        try:
            H.set_option("posterior_block", 1024)
            kx = H.empty(npad, Pp); mean = H.empty(P, 1); var = H.empty(P); S = H.empty(Pp, Pp)
            H.posterior(0, xd, theta, KV, alpha, 1, H.to_device(xp), kx, mean, var, S)
            H.sync()
        finally:
            H.set_option("posterior_block", 2048)
        assert np.max(np.abs(S.cpu().numpy()[:P, :P] - got[1][1])) < 1e-11 * theta[0]
        assert np.max(np.abs(S.cpu().numpy()[:P, :P] - want_S)) < 1e-10 * theta[0] + 1e-12
    Sfull = got[1][1]
    assert np.array_equal(Sfull, Sfull.T)          # the upper part is the mirror of the computed lower tiles
    for mode in (1, 0):
        m, S, v = got[mode]
        np.testing.assert_allclose(m, want_m, rtol=1e-8, atol=1e-9)
        assert np.max(np.abs(S - want_S)) < 1e-10 * theta[0] + 1e-12
        assert np.max(np.abs(v - np.diag(want_S))) < 1e-10 * theta[0] + 1e-12
    assert np.max(np.abs(got[1][1] - got[0][1])) < 1e-11 * theta[0]
