// Covariance assembly on gfx950: pairwise scaled distance + radial function, written once.
//
// Replaces get_anisotropic_distance_matrix / get_distance_matrix (fvgp/kernels.py:440-481)
// followed by squared_exponential_kernel / matern_kernel_diff1 / matern_kernel_diff2
// (kernels.py:16-33,98-118,166-188) and the K.copy()+fill_diagonal of GPkv.addKV
// (gp_kv.py:665-667), i.e. ~5d+6 N^2-sized numpy temporaries, by ONE streaming pass:
// the kernel is a pure HBM writer (8 B out per entry, 8*d B in per row/column).
//
// Layout: one 256-thread workgroup per 128x128 tile.  A wave owns whole rows: its 64 lanes
// hold two adjacent columns each, so every store instruction writes one contiguous 1 KiB
// row segment (16 B per lane).  The 128 row points sit in LDS (broadcast reads), the two
// column points in registers.  Tiles strictly above the diagonal are skipped in LOWER mode
// (the Cholesky never reads them), halving the bytes written.
//
// The same file holds the fused gradient kernel: sum_jk W_jk dK_jk/dtheta_i evaluated on
// the fly from x, replacing the (H,N,N) dK_dH tensor and the batched LU solve of
// gp_marginal_likelihood.py:260-274.
#include "common.h"

namespace {

constexpr double SQRT3 = 1.7320508075688772935;
constexpr double SQRT5 = 2.2360679774997896964;

struct KArgs {
    const double *x1; const double *x2; const double *vdiag; double *K;
    long n1, n2, ldk;
    int d, uplo, pad, vec_ok;
    double sig;
    double invl[FVGP_MAX_DIM];
};

// exp(-a) for a >= 0, a streaming kernel's version: Cody-Waite reduction a = n ln2 + r, |r| <= ln2 / 2, a degree-12 polynomial for
// exp(-r) (max. relative error 2e-17 of the polynomial, below 1 ulp with the rounding of the Horner steps), the scaling by one
// v_ldexp_f64 (which flushes through the subnormals to 0 by itself: no range checks).  17 fp64 operations, no comparison, no
// select (the library call carries four of each for arguments that cannot occur here).
__device__ __forceinline__ double exp_neg(const double a0) {
    // beyond 800 the result is 0 whatever the argument (+inf included: the reduction below would make inf - inf of it); a NaN
    // fails the comparison and stays a NaN, as numpy's exp leaves it (kernels.py:16-33)
    const double a = a0 > 800.0 ? 800.0 : a0;
    const double n = __builtin_rint(a * 1.4426950408889634074);          // a / ln 2
    double r = fma(n, -6.93147180369123816490e-01, a);                    // ln2 in two pieces: r = a - n ln2, exact product
    r = fma(n, -1.90821492927058770002e-10, r);
    const double x = -r;
    double p = 2.08767569878680989792e-09;                                // 1 / 12!
    p = fma(p, x, 2.50521083854417187751e-08);
    p = fma(p, x, 2.75573192239858906526e-07);
    p = fma(p, x, 2.75573192239858906526e-06);
    p = fma(p, x, 2.48015873015873015873e-05);
    p = fma(p, x, 1.98412698412698412698e-04);
    p = fma(p, x, 1.38888888888888888889e-03);
    p = fma(p, x, 8.33333333333333333333e-03);
    p = fma(p, x, 4.16666666666666666667e-02);
    p = fma(p, x, 1.66666666666666666667e-01);
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    p = fma(p, x, 1.0);
    return __builtin_ldexp(p, -(int)n);
}

// sqrt(x) for x >= 0 from the hardware reciprocal square root: one coupled Newton (Goldschmidt) step on g ~ sqrt(x), h ~ 1 / (2 sqrt(x))
// and one correction of g (the library's sqrt rescales for subnormal arguments, classifies its input and corrects twice: squared
// scaled distances need none of it).  x is first raised to 1e-300, so the diagonal (x = 0) gives 1e-150, which every radial function
// here maps to the same bits as 0.
__device__ __forceinline__ double sqrt_pos(const double x0) {
    const double x = x0 < 1e-300 ? 1e-300 : x0;          // (not fmax: a NaN distance stays a NaN, kernels.py:461-481)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

template <int KIND>
__device__ __forceinline__ double radial(double r2, double sig) {
    if (KIND == 0) return sig * exp_neg(0.5 * r2);
    const double r = sqrt_pos(r2);
    if (KIND == 1) { const double a = SQRT3 * r; return sig * (1.0 + a) * exp_neg(a); }
    const double a = SQRT5 * r;
    return sig * fma(5.0 / 3.0, r2, 1.0 + a) * exp_neg(a);
}

template <int KIND, int D>   // D == 0: runtime dimension (<= FVGP_MAX_DIM)
__global__ __launch_bounds__(256) void kmat_kernel(KArgs a) {
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (a.uplo == FVGP_LOWER && tj > ti) return;
    constexpr int DD = D ? D : FVGP_MAX_DIM;
    const int d = D ? D : a.d;
    __shared__ double sx[128 * DD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)ti * 128, col0 = (long)tj * 128;

    // stage the 128 row points (clamped: padding rows read a valid point and discard it)
    for (int e = tid; e < 128 * d; e += 256) {
        int rr = e / d, kk = e - rr * d;
        long gr = row0 + rr; if (gr >= a.n1) gr = a.n1 - 1;
        sx[rr * DD + kk] = a.x1[gr * d + kk];
    }
    const long c0 = col0 + 2 * lane, c1 = c0 + 1;
    double u0[DD], u1[DD], il[DD];
    {
        long g0 = c0 < a.n2 ? c0 : a.n2 - 1, g1 = c1 < a.n2 ? c1 : a.n2 - 1;
#pragma unroll
        for (int k = 0; k < DD; ++k) {
            if (k < d) { u0[k] = a.x2[g0 * d + k]; u1[k] = a.x2[g1 * d + k]; il[k] = a.invl[k]; }
            else { u0[k] = 0.0; u1[k] = 0.0; il[k] = 0.0; }
        }
    }
    __syncthreads();

    const bool ok0 = c0 < a.n2, ok1 = c1 < a.n2;
    // interior tiles (every row and column a real point, no diagonal entry): nothing but distance, radial function, store --
    // the per-entry tests for padding, identity and the noise on the diagonal cost a third of the instructions of an entry
    if (row0 + 128 <= a.n1 && col0 + 128 <= a.n2 && a.vec_ok && !(ti == tj && (a.vdiag != nullptr || a.pad == 1)) && row0 != col0) {
        double *dst = a.K + (row0 + wave) * a.ldk + c0;
        const long step = 4 * a.ldk;
        for (int rb = wave; rb < 128; rb += 8, dst += 2 * step) {
            double v[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rr = rb + 4 * u;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < DD; ++k) {
                    if (k < d) {
                        const double xr = sx[rr * DD + k];
                        const double e0 = (xr - u0[k]) * il[k], e1 = (xr - u1[k]) * il[k];
                        s0 = fma(e0, e0, s0); s1 = fma(e1, e1, s1);
                    }
                }
                v[u][0] = radial<KIND>(s0, a.sig); v[u][1] = radial<KIND>(s1, a.sig);
            }
            __builtin_nontemporal_store((double2_t){v[0][0], v[0][1]}, reinterpret_cast<double2_t *>(dst));
            __builtin_nontemporal_store((double2_t){v[1][0], v[1][1]}, reinterpret_cast<double2_t *>(dst + step));
        }
        return;
    }
    // two rows per trip: the second row's exp chain fills the latency of the first, and the stores go out
    // non-temporal (the matrix is written once and next read by another kernel: no reason to keep it in L2)
    for (int rb = wave; rb < 128; rb += 8) {
        double v[2][2];
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = rb + 4 * u;
            const long row = row0 + rr;
            const bool rok = row < a.n1;
            live[u] = rok || a.pad;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < DD; ++k) {
                if (k < d) {
                    const double xr = sx[rr * DD + k];
                    const double e0 = (xr - u0[k]) * il[k], e1 = (xr - u1[k]) * il[k];
                    s0 = fma(e0, e0, s0); s1 = fma(e1, e1, s1);
                }
            }
            double v0 = radial<KIND>(s0, a.sig), v1 = radial<KIND>(s1, a.sig);
            if (!(rok && ok0)) v0 = (a.pad == 1 && row == c0) ? 1.0 : 0.0;
            if (!(rok && ok1)) v1 = (a.pad == 1 && row == c1) ? 1.0 : 0.0;
            if (a.vdiag != nullptr && rok) {
                if (row == c0 && ok0) v0 += a.vdiag[row];
                if (row == c1 && ok1) v1 += a.vdiag[row];
            }
            v[u][0] = v0; v[u][1] = v1;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            double *dst = a.K + (row0 + rb + 4 * u) * a.ldk + c0;
            if (a.pad || (ok0 && ok1)) {
                if (a.vec_ok) __builtin_nontemporal_store((double2_t){v[u][0], v[u][1]}, reinterpret_cast<double2_t *>(dst));
                else { __builtin_nontemporal_store(v[u][0], dst); __builtin_nontemporal_store(v[u][1], dst + 1); }
            } else if (ok0) {
                __builtin_nontemporal_store(v[u][0], dst);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// fused gradient trace:  partial[block][i] = sum over the block's tile of
//      w_jk * (W_jk - b_j b_k) * dK_jk/dtheta_i ,   w = 1 on the diagonal, 2 below it
// (W symmetric, only its lower triangle is read).  theta = [sig, l_1..l_d] or [sig, l].
//   d/dsig = phi(r)                                       (all kinds)
//   rbf   : d/dl_i = k * D_i^2 / l_i^3                    (derived)
//   m32   : d/dl_i = sig * 3 * D_i^2 / l_i^3 * exp(-sqrt3 r)   (gp_prior.py:421-436, kernels.py:121-141)
//   m52   : d/dl_i = (5/3) sig (1+sqrt5 r) exp(-sqrt5 r) D_i^2 / l_i^3   (gp_bo.py:167-201)
// isotropic kinds sum the per-dimension terms into one d/dl.
struct GArgs {
    const double *x; const double *W; const double *b; double *partial;
    long n, ldw, ldb;
    int d, iso, ntheta;
    double sig;
    double invl[FVGP_MAX_DIM];
    int ntj, tj0;             // column-slab mode (ntj > 0): W holds the tile columns tj0 .. tj0 + ntj - 1 only (its column 0 is
    long wcol0;               //   matrix column wcol0); grid = (tile rows, ntj)
};

// phi and the common factor cf such that dK/dl_k = cf * e2[k] * invl[k]  (e2 = D^2 / l^2), one exp and one square root per entry
template <int KIND>
__device__ __forceinline__ void radial_grad(const double r2, const double sig, double &phi, double &cf) {
    if (KIND == 0) { phi = exp_neg(0.5 * r2); cf = sig * phi; return; }
    const double r = sqrt_pos(r2);
    if (KIND == 1) { const double ea = exp_neg(SQRT3 * r); phi = fma(SQRT3, r, 1.0) * ea; cf = 3.0 * sig * ea; return; }
    const double ea = exp_neg(SQRT5 * r), t = fma(SQRT5, r, 1.0);
    phi = fma(5.0 / 3.0, r2, t) * ea; cf = (5.0 / 3.0) * sig * t * ea;
}

template <int KIND, int D>   // D == 0: runtime dimension (<= FVGP_MAX_DIM); a 16-deep predicated loop per entry made d = 3 run at a tenth of the memory rate
__global__ __launch_bounds__(256) void grad_trace_kernel(GArgs a) {
    int ti, tj;
    long pidx = blockIdx.x;
    if (a.ntj > 0) {          // slab: one block per (tile row, tile column of the window); tiles above the diagonal hold nothing
        ti = blockIdx.x; tj = a.tj0 + blockIdx.y;
        pidx = (long)blockIdx.y * gridDim.x + blockIdx.x;
        if (ti < tj) {
            if (threadIdx.x < a.ntheta) a.partial[pidx * a.ntheta + threadIdx.x] = 0.0;
            return;
        }
    } else {                  // blockIdx.x enumerates lower-triangular tiles
        const long t = blockIdx.x;
        ti = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while ((long)ti * (ti + 1) / 2 > t) --ti;
        tj = (int)(t - (long)ti * (ti + 1) / 2);
    }

    constexpr int DD = D ? D : FVGP_MAX_DIM;
    const int d = D ? D : a.d;
    __shared__ double sx[128 * DD];
    __shared__ double sb[128];
    __shared__ double sred[4][DD + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)ti * 128, col0 = (long)tj * 128;
    for (int e = tid; e < 128 * d; e += 256) {
        int rr = e / d, kk = e - rr * d;
        long gr = row0 + rr; if (gr >= a.n) gr = a.n - 1;
        sx[rr * DD + kk] = a.x[gr * d + kk];
    }
    if (tid < 128) { long gr = row0 + tid; sb[tid] = (a.b && gr < a.n) ? a.b[gr * a.ldb] : 0.0; }
    const long c0 = col0 + 2 * lane, c1 = c0 + 1;
    double u0[DD], u1[DD], il[DD];
    const long g0 = c0 < a.n ? c0 : a.n - 1, g1 = c1 < a.n ? c1 : a.n - 1;
#pragma unroll
    for (int k = 0; k < DD; ++k) {
        if (k < d) { u0[k] = a.x[g0 * d + k]; u1[k] = a.x[g1 * d + k]; il[k] = a.invl[k]; }
        else { u0[k] = 0.0; u1[k] = 0.0; il[k] = 0.0; }
    }
    const double bc0 = (a.b && c0 < a.n) ? a.b[c0 * a.ldb] : 0.0, bc1 = (a.b && c1 < a.n) ? a.b[c1 * a.ldb] : 0.0;
    __syncthreads();

    double gs = 0.0;          // d/dsig accumulator
    double gl[DD];            // d/dl_k accumulators, WITHOUT the factor 1 / l_k (applied once at the end)
#pragma unroll
    for (int k = 0; k < DD; ++k) gl[k] = 0.0;

    // one entry: weight wt (0 for entries that do not count), row point rr, column point h
    auto entry = [&](const int rr, const int h, const double wt) {
        double e2[DD];
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < DD; ++k) {
            if (k < d) {
                const double e = (sx[rr * DD + k] - (h ? u1[k] : u0[k])) * il[k];
                e2[k] = e * e; r2 += e2[k];
            } else e2[k] = 0.0;
        }
        double phi, cf;
        radial_grad<KIND>(r2, a.sig, phi, cf);
        gs = fma(wt, phi, gs);
        const double wc = wt * cf;
#pragma unroll
        for (int k = 0; k < DD; ++k) if (k < d) gl[k] = fma(wc, e2[k], gl[k]);
    };
    const double *Wp = a.W + (row0 + wave) * a.ldw + (c0 - a.wcol0);
    if (row0 + 128 <= a.n && ti != tj) {
        // interior tile strictly below the diagonal: every entry counts twice, two rows per trip (their loads of W and their
        // exp / rsq chains interleave)
        for (int rr = wave; rr < 128; rr += 8, Wp += 8 * a.ldw) {
            const double2_t wa = *reinterpret_cast<const double2_t *>(Wp), wb = *reinterpret_cast<const double2_t *>(Wp + 4 * a.ldw);
            const double bra = sb[rr], brb = sb[rr + 4];
            entry(rr, 0, 2.0 * (wa[0] - bra * bc0));
            entry(rr, 1, 2.0 * (wa[1] - bra * bc1));
            entry(rr + 4, 0, 2.0 * (wb[0] - brb * bc0));
            entry(rr + 4, 1, 2.0 * (wb[1] - brb * bc1));
        }
    } else {
        for (int rr = wave; rr < 128; rr += 4, Wp += 4 * a.ldw) {
            const long row = row0 + rr;
            if (row >= a.n) break;
            const double2_t w2 = *reinterpret_cast<const double2_t *>(Wp);
            const double br = sb[rr];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long c = h ? c1 : c0;
                if (c > row || c >= a.n) continue;
                entry(rr, h, (c == row ? 1.0 : 2.0) * ((h ? w2[1] : w2[0]) - br * (h ? bc1 : bc0)));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DD; ++k) gl[k] *= il[k];
    // wave reduce then block reduce
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gs += __shfl_down(gs, off, 64);
#pragma unroll
        for (int k = 0; k < DD; ++k) if (k < d) gl[k] += __shfl_down(gl[k], off, 64);
    }
    if (lane == 0) {
        sred[wave][0] = gs;
#pragma unroll
        for (int k = 0; k < DD; ++k) if (k < d) sred[wave][1 + k] = gl[k];
    }
    __syncthreads();
    if (tid == 0) {
        double *out = a.partial + pidx * a.ntheta;
        double s = sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0];
        out[0] = s;
        if (a.iso) {
            double acc = 0.0;
            for (int k = 0; k < d; ++k) acc += sred[0][1 + k] + sred[1][1 + k] + sred[2][1 + k] + sred[3][1 + k];
            out[1] = acc;
        } else {
            for (int k = 0; k < d; ++k) out[1 + k] = sred[0][1 + k] + sred[1][1 + k] + sred[2][1 + k] + sred[3][1 + k];
        }
    }
}

}  // namespace

int kmat_desc_from_theta(int kernel_id, int d, const double *theta, int ntheta, KmatDesc *out) {
    if (kernel_id < 0 || kernel_id > 5) { fvgp_set_error("unknown kernel id"); return -2; }
    if (d < 1 || d > FVGP_MAX_DIM) { fvgp_set_error("input dimension out of range"); return -7; }
    const bool iso = kernel_id >= 3;
    if (ntheta < (iso ? 2 : d + 1)) { fvgp_set_error("too few hyperparameters for this kernel"); return -9; }
    out->kind = kernel_id % 3;
    out->d = d;
    out->sig = theta[0];
    for (int k = 0; k < FVGP_MAX_DIM; ++k) out->invl[k] = 0.0;
    for (int k = 0; k < d; ++k) out->invl[k] = 1.0 / (iso ? theta[1] : theta[1 + k]);
    return 0;
}

int launch_kmat(fvgp_handle *h, const KmatDesc &k) {
    if (k.n1 <= 0 || k.n2 <= 0) return 0;
    KArgs a;
    a.x1 = k.x1; a.x2 = k.x2; a.vdiag = k.vdiag; a.K = k.K;
    a.n1 = k.n1; a.n2 = k.n2; a.ldk = k.ldk; a.d = k.d; a.uplo = k.uplo; a.pad = k.pad;
    a.sig = k.sig;
    for (int i = 0; i < FVGP_MAX_DIM; ++i) a.invl[i] = k.invl[i];
    a.vec_ok = ((k.ldk & 1) == 0 && ((uintptr_t)k.K & 15) == 0) ? 1 : 0;
    dim3 grid((unsigned)((k.n2 + 127) / 128), (unsigned)((k.n1 + 127) / 128)), block(256);
#define GO(KIND, D) hipLaunchKernelGGL((kmat_kernel<KIND, D>), grid, block, 0, h->stream, a)
#define GOD(KIND)                                   \
    switch (k.d) {                                  \
        case 1: GO(KIND, 1); break;                 \
        case 2: GO(KIND, 2); break;                 \
        case 3: GO(KIND, 3); break;                 \
        case 4: GO(KIND, 4); break;                 \
        default: GO(KIND, 0); break;                \
    }
    switch (k.kind) {
        case 0: GOD(0); break;
        case 1: GOD(1); break;
        default: GOD(2); break;
    }
#undef GOD
#undef GO
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_grad_trace(fvgp_handle *h, const GradDesc &g, int *nblocks_out) {
    GArgs a;
    a.x = g.k.x1; a.W = g.W; a.b = g.b; a.partial = g.partial;
    a.n = g.k.n1; a.ldw = g.ldw; a.ldb = g.ldb; a.d = g.k.d; a.iso = g.kernel_id >= 3; a.ntheta = g.ntheta;
    a.sig = g.k.sig;
    for (int i = 0; i < FVGP_MAX_DIM; ++i) a.invl[i] = g.k.invl[i];
    const long T = (a.n + 127) / 128;
    a.ntj = 0; a.tj0 = 0; a.wcol0 = 0;
    long nb = T * (T + 1) / 2;
    dim3 grid((unsigned)nb), block(256);
    if (g.ncols > 0) {        // a slab of columns [col0, col0 + ncols) of the symmetric matrix
        a.tj0 = (int)(g.col0 / 128); a.ntj = (int)((g.ncols + 127) / 128); a.wcol0 = g.col0;
        if (a.tj0 + a.ntj > T) a.ntj = (int)(T - a.tj0);
        if (a.ntj <= 0) { *nblocks_out = 0; return 0; }
        grid = dim3((unsigned)T, (unsigned)a.ntj);
        nb = T * a.ntj;
    }
    *nblocks_out = (int)nb;
#define GT(KIND, D) hipLaunchKernelGGL((grad_trace_kernel<KIND, D>), grid, block, 0, h->stream, a)
#define GTD(KIND)                                   \
    switch (a.d) {                                  \
        case 1: GT(KIND, 1); break;                 \
        case 2: GT(KIND, 2); break;                 \
        case 3: GT(KIND, 3); break;                 \
        case 4: GT(KIND, 4); break;                 \
        default: GT(KIND, 0); break;                \
    }
    switch (g.k.kind) {
        case 0: GTD(0); break;
        case 1: GTD(1); break;
        default: GTD(2); break;
    }
#undef GTD
#undef GT
    HIPCHK(hipGetLastError());
    return 0;
}
