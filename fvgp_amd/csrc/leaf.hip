// 128x128 diagonal-block Cholesky + triangular inverse, one workgroup, all in LDS.
//
// Stands in for the unblocked dpotf2 / dtrtri steps inside LAPACK dpotrf that
// scipy.linalg.cho_factor reaches (fvgp/gp_lin_alg.py:245).  The blocked driver
// (potrf.hip) calls it once per 128 columns; its outputs are
//   * L11 in place (lower triangle of the block only -- the strict upper is never read or
//     written, matching cho_factor's "upper is unspecified"),
//   * inv(L11) as a dense 128x128 (upper = 0) so that the panel TRSM  A21 * L11^-T  and the
//     block substitutions of potrs become plain MFMA GEMMs,
//   * sum(log L_ii) of the block (feeds calculate_Chol_logdet, gp_lin_alg.py:337-338),
//   * info: 1-based global index of the first non-positive pivot (dpotrf's info > 0).
//
// Algorithm (16-wide blocks, fp64 MFMA 16x16x4 for every tile product):
//   for p = 0..7:  wave 0 factors the 16x16 diagonal tile with lane-shuffles (row per lane,
//                  left-looking) and inverts it (column per lane);  all waves: tile TRSM
//                  X = A * Dinv^T;  trailing tiles C -= X_i X_j^T.
//   inverse:       block column j of inv(L) by forward block substitution
//                  X_ij = -Dinv_i * sum_{k=j..i-1} L_ik X_kj ; the MFMA accumulator layout
//                  (row = q+4v) is exactly the next MFMA's B-operand layout, so the product
//                  chains in registers.  X is kept transposed in the (otherwise unused)
//                  upper triangle of the LDS image.
#include "common.h"

namespace {

constexpr int LS = 130;   // LDS row stride (doubles): 260 dwords == 4 mod 64 -> conflict-free fragment reads
constexpr int DS = 18;    // stride of the 16x16 diagonal-inverse tiles

struct LeafArgs {
    double *A; long lda;          // block origin
    double *linv;                 // 128*128 out
    double *logdet_part;          // 1 double out (may be null)
    int *info; int info_base;
    int do_factor;
    int nvalid;                   // rows of this block that count for log-det and info (rest is padding)
    long a_stride, linv_stride;   // batched mode: block b at A + b*a_stride
};

__device__ __forceinline__ double4_t mfma(double a, double b, double4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// one wave: factor (optional) and invert the 16x16 tile at sT (stride LS); inverse -> sDt (stride DS)
// returns the first bad pivot column (0..15) or -1
template <bool do_factor>
__device__ __forceinline__ int diag_tile(double *sT, double *sDt, int lane) {
    const int row = lane & 15;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (k <= row) ? sT[row * LS + k] : 0.0;
    int bad = -1;
    if (do_factor) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double s = a[j];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < j) s -= a[k] * __shfl(a[k], j, 64);
            const double dj = __shfl(s, j, 64);
            if (!(dj > 0.0) && bad < 0) bad = j;
            const double piv = sqrt(dj);
            a[j] = (row == j) ? piv : (row > j ? s / piv : 0.0);
        }
    }
    double x[16];
    const int c = row;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < i) s -= __shfl(a[k], i, 64) * x[k];
        x[i] = s / __shfl(a[i], i, 64);
    }
    if (lane < 16) {
        if (do_factor) {
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k <= row) sT[row * LS + k] = a[k];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sDt[i * DS + c] = x[i];
    }
    return bad;
}

__global__ __launch_bounds__(256) void leaf_kernel(LeafArgs g) {
    __shared__ double sA[128 * LS];
    __shared__ double sD[8 * 16 * DS];
    __shared__ double slog[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    double *A = g.A + (long)blockIdx.x * g.a_stride;
    double *linv = g.linv + (long)blockIdx.x * g.linv_stride;

    // load the lower triangle; zero the strict upper (it becomes the workspace for inv(L)^T)
    for (int e = tid; e < 128 * 64; e += 256) {
        const int row = e >> 6, c2 = (e & 63) * 2;
        double2_t v = {0.0, 0.0};
        if (c2 <= row) v = *reinterpret_cast<const double2_t *>(A + (long)row * g.lda + c2);
        if (c2 + 1 > row) v[1] = 0.0;
        if (c2 > row) v[0] = 0.0;
        sA[row * LS + c2] = v[0];
        sA[row * LS + c2 + 1] = v[1];
    }
    __syncthreads();

    if (g.do_factor) {
        for (int p = 0; p < 8; ++p) {
            if (wave == 0) {
                const int bad = diag_tile<true>(&sA[(16 * p) * LS + 16 * p], &sD[p * 16 * DS], lane);
                if (bad >= 0 && lane == 0 && 16 * p + bad < g.nvalid) atomicCAS(g.info, 0, g.info_base + (int)blockIdx.x * 128 + 16 * p + bad + 1);
            }
            __syncthreads();
            // TRSM: X_t = A_t * Dinv_p^T for tiles t = p+1..7
            for (int t = p + 1 + wave; t < 8; t += 4) {
                double a[4], b[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    a[s] = sA[(16 * t + r) * LS + 16 * p + 4 * s + q];
                    b[s] = sD[p * 16 * DS + r * DS + 4 * s + q];
                }
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma(a[s], b[s], acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) sA[(16 * t + q + 4 * v) * LS + 16 * p + r] = acc[v];
            }
            __syncthreads();
            // trailing update: C_ij -= X_i X_j^T for p < j <= i <= 7
            const int T = 7 - p;
            const int ntile = T * (T + 1) / 2;
            for (int idx = wave; idx < ntile; idx += 4) {
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= idx) ++ii;
                const int jj = idx - ii * (ii + 1) / 2;
                const int i = p + 1 + ii, j = p + 1 + jj;
                double4_t acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = sA[(16 * i + q + 4 * v) * LS + 16 * j + r];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = -sA[(16 * i + r) * LS + 16 * p + 4 * s + q];
                    const double b = sA[(16 * j + r) * LS + 16 * p + 4 * s + q];
                    acc = mfma(a, b, acc);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) sA[(16 * i + q + 4 * v) * LS + 16 * j + r] = acc[v];
            }
            __syncthreads();
        }
        // the trailing updates also touched the strict upper part of diagonal tiles; re-zero it
        for (int e = tid; e < 8 * 256; e += 256) {
            const int t = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            if (cc > rr) sA[(16 * t + rr) * LS + 16 * t + cc] = 0.0;
        }
        __syncthreads();
    } else {
        // inverse only: invert the 8 diagonal tiles of the given factor
        for (int p = wave; p < 8; p += 4) diag_tile<false>(&sA[(16 * p) * LS + 16 * p], &sD[p * 16 * DS], lane);
        __syncthreads();
    }

    // write L back (lower triangle only) and its log-diagonal sum
    if (g.do_factor) {
        for (int e = tid; e < 128 * 128; e += 256) {
            const int row = e >> 7, col = e & 127;
            if (col <= row) A[(long)row * g.lda + col] = sA[row * LS + col];
        }
        if (g.logdet_part != nullptr) {
            double s = 0.0;
            if (tid < g.nvalid) s = log(fabs(sA[tid * LS + tid]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) slog[wave] = s;
            __syncthreads();
            if (tid == 0) g.logdet_part[blockIdx.x] = slog[0] + slog[1];
        }
    }

    // inv(L): block column j, block rows i = j+1..7, X_ij^T stored at tile (j,i)
    for (int i = 1; i < 8; ++i) {
        for (int j = wave; j < i; j += 4) {
            double4_t t4 = {0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = sA[(16 * i + r) * LS + 16 * k + 4 * s + q];              // L_ik[r][4s+q]
                    const double b = (k == j) ? sD[j * 16 * DS + (4 * s + q) * DS + r]          // Dinv_j[4s+q][r]
                                              : sA[(16 * j + r) * LS + 16 * k + 4 * s + q];     // X_kj[4s+q][r] (transposed store)
                    t4 = mfma(a, b, t4);
                }
            }
            double4_t x4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -sD[i * 16 * DS + r * DS + 4 * s + q];                        // -Dinv_i[r][4s+q]
                x4 = mfma(a, t4[s], x4);                                                       // T[4s+q][r] == acc register s
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) sA[(16 * j + r) * LS + 16 * i + q + 4 * v] = x4[v];    // X_ij[q+4v][r] -> transposed
        }
        __syncthreads();
    }

    // write inv(L) (dense, upper = 0)
    for (int e = tid; e < 128 * 128; e += 256) {
        const int row = e >> 7, col = e & 127;
        const int tr = row >> 4, tc = col >> 4;
        double v = 0.0;
        if (tr == tc) v = sD[tr * 16 * DS + (row & 15) * DS + (col & 15)];
        else if (tr > tc) v = sA[col * LS + row];
        linv[(long)row * 128 + col] = v;
    }
}

}  // namespace

int launch_leaf(fvgp_handle *h, double *A, int64_t lda, double *linv, double *logdet_part, int info_base, int do_factor, int nvalid) {
    LeafArgs g;
    g.A = A; g.lda = lda; g.linv = linv; g.logdet_part = logdet_part; g.info = h->dinfo; g.info_base = info_base;
    g.do_factor = do_factor; g.a_stride = 0; g.linv_stride = 0; g.nvalid = nvalid;
    hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_leaf_inverse_batched(fvgp_handle *h, const double *L, int64_t ldl, int64_t nblk, double *linv) {
    if (nblk <= 0) return 0;
    LeafArgs g;
    g.A = const_cast<double *>(L); g.lda = ldl; g.linv = linv; g.logdet_part = nullptr; g.info = h->dinfo; g.info_base = 0;
    g.do_factor = 0; g.a_stride = 128 * ldl + 128; g.linv_stride = LEAF_DOUBLES; g.nvalid = 128;
    hipLaunchKernelGGL(leaf_kernel, dim3((unsigned)nblk), dim3(256), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}
