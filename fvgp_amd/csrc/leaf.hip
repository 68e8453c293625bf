// 128x128 diagonal-block Cholesky + triangular inverse, one workgroup, all in LDS.
//
// Stands in for the unblocked dpotf2 / dtrtri steps inside LAPACK dpotrf that
// scipy.linalg.cho_factor reaches (fvgp/gp_lin_alg.py:245).  The blocked driver
// (api.hip: panel_factor) calls it once per 128 columns; its outputs are
//   * L11 in place (lower triangle of the block only -- the strict upper is never read or
//     written, matching cho_factor's "upper is unspecified"),
//   * inv(L11) as a dense 128x128 (upper = 0) so that the panel TRSM  A21 * L11^-T  and the
//     block substitutions of potrs become plain MFMA GEMMs,
//   * sum(log L_ii) of the block's valid rows (feeds calculate_Chol_logdet, gp_lin_alg.py:337-338),
//   * info: 1-based global index of the first non-positive pivot (dpotrf's info > 0).
//
// Resource shape matters as much as speed: under look-ahead this kernel must start while the
// trailing SYRK occupies every CU with two 72 KB workgroups, so it is built to fit into what ONE
// retiring SYRK workgroup frees: the lower triangle is held as 36 packed 16x16 tiles
// (36 x 2048 B = 72 KB + 1 KB, below the 76 KB a SYRK workgroup holds), 512 threads at <= 128 VGPRs.
//
// Algorithm (16-wide blocks):
//   for p = 0..7:  TRSM of the rows below by per-row forward substitution against L_pp;
//                  trailing tiles C_ij -= X_i X_j^T on fp64 MFMA 16x16x4; wave 0 takes tile
//                  (p+1,p+1) first and factors it while the other waves finish the update: four-column
//                  panels whose 4x4 diagonal block is factored as wave-uniform scalars (Newton reciprocal
//                  square roots), the rows under it solved per lane group, and ONE rank-4 MFMA update of the
//                  rest of the tile per panel (PanelBlock).
//   inverse:       wave j owns block column j of inv(L):  X_jj = inv(L_jj),
//                  X_ij = -inv(L_ii) * sum_{k=j..i-1} L_ik X_kj.  The f64 MFMA accumulator layout
//                  (row = q+4v) is exactly the next MFMA's B-operand layout, so the whole column
//                  stays in registers and goes straight to global memory.
#include "common.h"

namespace {

constexpr int TSZ = 256;          // a packed 16x16 tile, unpadded: 36 tiles = exactly 72 KB
constexpr int NT = 36;            // tiles (i,j), j <= i < 8

// element (a, b) of a tile: columns XOR-swizzled by an even mask so that both MFMA fragment read
// patterns (16 rows x one column pair, one row pair x 16 columns) are LDS bank-conflict free
__device__ __forceinline__ int el(int a, int b) { return a * 16 + (b ^ (((a >> 1) & 7) << 1)); }

struct LeafArgs {
    double *A; long lda;          // block origin
    double *linv;                 // 128*128 out
    double *logdet_part;          // 128 doubles out: 1 / L_ii of the block's valid rows, 1 for padding (may be null)
    int *info; int info_base;
    int do_factor;
    int nvalid;                   // rows of this block that count for log-det and info (rest is padding)
    long a_stride, linv_stride;   // batched mode: block b at A + b*a_stride
    unsigned long *stamps;        // diagnostics: s_memtime at the phase boundaries (nullptr in the product path)
    int tiles_only;               // linv <- the inverses of the eight 16x16 diagonal tiles only (8 x 256 doubles, lower, zeros above)
    int *yield;                   // per-CU counters the trailing update's waves poll (common.h, cu_yield); nullptr: nobody yields
};

__device__ __forceinline__ double4_t mfma(double a, double b, double4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int tix(int i, int j) { return (i * (i + 1) / 2 + j) * TSZ; }

// value of `v` in lane SRC (compile-time constant) broadcast to the whole wave through SGPRs
template <int SRC>
__device__ __forceinline__ double bcast(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to fp64 accuracy: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

template <int I>
struct InvRow {
    // row I of inv(tile): lane c holds column c of the inverse in x[]
    static __device__ __forceinline__ void step(const double (&a)[16], const double (&rd)[16], double (&x)[16], int c) {
        double s = (I == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < I; ++k) s = fma(-bcast<I>(a[k]), x[k], s);
        x[I] = s * rd[I];
        if constexpr (I < 15) InvRow<I + 1>::step(a, rd, x, c);
    }
};

// the 16-lane block BLK of v in all four blocks of the wave (two gfx950 row swaps per 32-bit half)
template <int BLK>
__device__ __forceinline__ double bcast_block(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // [0] = blocks (0,0,2,2), [1] = (1,1,3,3)
    const auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned l = (BLK & 1) ? l1[1] : l1[0], h = (BLK & 1) ? h1[1] : h1[0];
    const auto l2 = __builtin_amdgcn_permlane32_swap(l, l, false, false);        // [0] = lower half twice, [1] = upper half twice
    const auto h2 = __builtin_amdgcn_permlane32_swap(h, h, false, false);
    return __hiloint2double((BLK & 2) ? h2[1] : h2[0], (BLK & 2) ? l2[1] : l2[0]);
}

// sqrt(d) from d and y ~ 1/sqrt(d): one correction step on top of d y
__device__ __forceinline__ double sqrt_from(double d, double y) {
    const double p = d * y;
    return fma(fma(-p, p, d), 0.5 * y, p);
}

// Four-column block B of a 16x16 tile.  The tile is kept as the full symmetric matrix in the MFMA accumulator layout, lane
// (r, q) holds D[q + 4v][r] = D[r][q + 4v], so accumulator slot B of lane (r, q) IS the panel entry (row r, panel column q).
//   1. the 4x4 diagonal block is factored as wave-uniform scalars (its ten entries by readlane): the pivot chain -- four
//      reciprocal square roots in sequence -- waits for nothing else;
//   2. every row of the panel is solved against it, column c in lane group c; a finished column reaches the other lane
//      groups by two gfx950 row swaps per 32-bit half (bcast_block);
//   3. the factored panel is both operands of ONE rank-4 MFMA update of the rest of the tile.
// (The first version swept 16 columns with 15 - J broadcast-and-FMA updates behind each pivot: 7.4k cycles per tile.)
template <int B>
struct PanelBlock {
    static __device__ __forceinline__ void step(double4_t &acc, double *sT, double &ykeep, int r, int q, int lane, int &bad) {
        constexpr int R0 = 4 * B;
        const bool above = r < R0 + q;                        // finished rows, and the 4x4 block above its diagonal
        const double x = above ? 0.0 : acc[B];
        const double d00 = bcast<R0>(x), d10 = bcast<R0 + 1>(x), d20 = bcast<R0 + 2>(x), d30 = bcast<R0 + 3>(x);
        const double d11 = bcast<16 + R0 + 1>(x), d21 = bcast<16 + R0 + 2>(x), d31 = bcast<16 + R0 + 3>(x);
        const double d22 = bcast<32 + R0 + 2>(x), d32 = bcast<32 + R0 + 3>(x);
        const double d33 = bcast<48 + R0 + 3>(x);
        if (!(d00 > 0.0) && bad < 0) bad = R0;
        const double y0 = rsqrt_nr(d00);
        const double l10 = d10 * y0, l20 = d20 * y0, l30 = d30 * y0;
        const double e11 = fma(-l10, l10, d11);
        if (!(e11 > 0.0) && bad < 0) bad = R0 + 1;
        const double y1 = rsqrt_nr(e11);
        const double l21 = fma(-l20, l10, d21) * y1, l31 = fma(-l30, l10, d31) * y1;
        const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
        if (!(e22 > 0.0) && bad < 0) bad = R0 + 2;
        const double y2 = rsqrt_nr(e22);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * y2;
        const double e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        if (!(e33 > 0.0) && bad < 0) bad = R0 + 3;
        const double y3 = rsqrt_nr(e33);
        // rows of the panel: x_c = (raw_c - sum_{k<c} x_k L[c][k]) y_c, valid in lane group c
        const double x0 = x * y0;
        const double X0 = bcast_block<0>(x0);
        const double x1 = fma(-X0, l10, x) * y1;
        const double X1 = bcast_block<1>(x1);
        const double x2 = fma(-X1, l21, fma(-X0, l20, x)) * y2;
        const double X2 = bcast_block<2>(x2);
        const double x3 = fma(-X2, l32, fma(-X1, l31, fma(-X0, l30, x))) * y3;
        const bool g0 = q == 0, g1 = q == 1, g2 = q == 2;
        double xf = g0 ? x0 : (g1 ? x1 : (g2 ? x2 : x3));
        // the diagonal entries with one correction step (on this lane group's pivot), zeros above them
        const double eq = g0 ? d00 : (g1 ? e11 : (g2 ? e22 : e33));
        const double yq = g0 ? y0 : (g1 ? y1 : (g2 ? y2 : y3));
        const double pd = sqrt_from(eq, yq);
        if (r == R0 + q) { xf = pd; ykeep = yq; }            // lane (r, q = r mod 4) keeps 1 / L_rr
        if (above) xf = 0.0;
        if (r >= R0 + q) sT[el(r, R0 + q)] = xf;
        if constexpr (B < 3) {
            acc = mfma(-xf, xf, acc);
            PanelBlock<B + 1>::step(acc, sT, ykeep, r, q, lane, bad);
        }
    }
};

// one wave: Cholesky of the 16x16 tile at sT (lower part), in place; 1/diag -> srd[0..15]
// returns the first bad pivot column or -1
__device__ __forceinline__ int diag_factor(double *sT, double *srd, int lane) {
    const int r = lane & 15, q = lane >> 4;
    double4_t acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int i = q + 4 * v; acc[v] = (i >= r) ? sT[el(i, r)] : sT[el(r, i)]; }
    int bad = -1;
    double ykeep = 0.0;
    PanelBlock<0>::step(acc, sT, ykeep, r, q, lane, bad);
    if (q == (r & 3)) srd[r] = ykeep;
    return bad;
}

__global__ __launch_bounds__(512, 4) void leaf_kernel(LeafArgs g) {
    __shared__ double sT[NT * TSZ];      // 73,728 B
    __shared__ double srd[128];          // 1 / L_aa
    // the leaf sits on the critical path of the panel chain and shares its SIMDs with trailing-update waves
    // (look-ahead): its instructions go first
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    double *A = g.A + (long)blockIdx.x * g.a_stride;
    double *linv = g.linv + (long)blockIdx.x * g.linv_stride;
    int nst = 0;
    // this compute unit is the leaf's while it runs: the co-resident trailing-update workgroup sleeps (gemm.hip, YIELD)
    int *const yslot = g.yield ? cu_yield_slot(g.yield) : nullptr;      // wave-uniform: lives in scalar registers
    if (yslot && tid == 0) atomicAdd(yslot, 1);
#define FVGP_STAMP() do { if (g.stamps && tid == 0) g.stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
    FVGP_STAMP();

    // ---- load the lower triangle into packed tiles; strict upper of diagonal tiles <- 0 ----------
    {
        // eight loads of a thread in flight per LDS-write batch (two memory round trips instead of sixteen)
        const int c2 = (tid & 63) * 2, tj = c2 >> 4, rbase = tid >> 6;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            double2_t v[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + 8 * (8 * hb + it);
                v[it] = (double2_t){0.0, 0.0};
                if (tj <= (row >> 4)) v[it] = *reinterpret_cast<const double2_t *>(A + (long)row * g.lda + c2);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + 8 * (8 * hb + it), ti = row >> 4;
                if (tj > ti) continue;
                if (c2 + 1 > row) v[it][1] = 0.0;
                if (c2 > row) v[it][0] = 0.0;
                double *dst = &sT[tix(ti, tj)];
                dst[el(row & 15, c2 & 15)] = v[it][0]; dst[el(row & 15, (c2 & 15) + 1)] = v[it][1];
            }
        }
    }
    __syncthreads();
    FVGP_STAMP();

    if (g.do_factor) {
        if (wave == 0) {
            const int bad = diag_factor(&sT[tix(0, 0)], &srd[0], lane);
            if (bad >= 0 && lane == 0 && bad < g.nvalid) atomicCAS(g.info, 0, g.info_base + (int)blockIdx.x * 128 + bad + 1);
        }
        __syncthreads();
        for (int p = 0; p < 8; ++p) {
            // ---- TRSM: rows below the diagonal tile, one thread per row, x <- a * L_pp^-T ------------
            const int R = 112 - 16 * p;
            if (tid < R) {
                const int i = p + 1 + (tid >> 4), a = tid & 15;
                double *rowp = &sT[tix(i, p)];
                const double *Lp = &sT[tix(p, p)];
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = rowp[el(a, c)];
                // column sweep: a finished x[j] is applied to all later entries at once (independent FMAs), so the dependent
                // path is 16 x (scale, one FMA) instead of a j-term dot product per entry
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j] *= srd[16 * p + j];
#pragma unroll
                    for (int c = 0; c < 16; ++c) if (c > j) x[c] = fma(-x[j], Lp[el(c, j)], x[c]);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) rowp[el(a, c)] = x[c];
            }
            __syncthreads();
            FVGP_STAMP();
            // ---- trailing update C_ij -= X_i X_j^T, p < j <= i <= 7 ---------------------------------------
            const int T = 7 - p;
            const int ntile = T * (T + 1) / 2;
            // wave 0: tile (p+1,p+1) only, then it factors that tile; waves 1..7 share the rest
            for (int idx = wave; idx < ntile; idx += (wave == 0 ? 1000 : 7)) {
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= idx) ++ii;
                const int jj = idx - ii * (ii + 1) / 2;
                const int i = p + 1 + ii, j = p + 1 + jj;
                double *C = &sT[tix(i, j)];
                const double *Xi = &sT[tix(i, p)], *Xj = &sT[tix(j, p)];
                double4_t acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = C[el(q + 4 * v, r)];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma(-Xi[el(r, 4 * s + q)], Xj[el(r, 4 * s + q)], acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) if (i != j || r <= q + 4 * v) C[el(q + 4 * v, r)] = acc[v];
            }
            FVGP_STAMP();
            if (wave == 0 && p < 7) {
                const int bad = diag_factor(&sT[tix(p + 1, p + 1)], &srd[16 * (p + 1)], lane);
                if (bad >= 0 && lane == 0 && 16 * (p + 1) + bad < g.nvalid)
                    atomicCAS(g.info, 0, g.info_base + (int)blockIdx.x * 128 + 16 * (p + 1) + bad + 1);
            }
            __syncthreads();
            FVGP_STAMP();
        }
        // ---- L back to global (lower triangle only), tile by tile: a wave's store covers four 128-byte row segments ----
        for (int t = wave; t < NT; t += 8) {
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            const double *T = &sT[t * TSZ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = 4 * u + q;
                if (ti != tj || r <= a) A[(long)(16 * ti + a) * g.lda + 16 * tj + r] = T[el(a, r)];
            }
        }
        // ---- the reciprocal diagonal, for the log-determinant: the logarithms are taken once, by one kernel over all blocks
        //      (neg_log_sum_kernel), not 128 at a time behind a barrier on the chain's critical path (1.4 thousand cycles) ------
        if (g.logdet_part != nullptr && tid < 128) g.logdet_part[(long)blockIdx.x * 128 + tid] = tid < g.nvalid ? srd[tid] : 1.0;
        FVGP_STAMP();
    } else {
        if (tid < 128) srd[tid] = 1.0 / sT[tix(tid >> 4, tid >> 4) + el(tid & 15, tid & 15)];
        __syncthreads();
    }

    // ---- inverse of the 8 diagonal tiles: wave w inverts tile (w,w); its strictly-lower part goes,
    //      transposed, into the tile's (unused) strict upper half, the diagonal is srd -------------------
    double x[16];
    {
        // lane c solves L x = e_c (column c of the inverse) by a COLUMN sweep: a finished x[k] goes into every later row at once
        // (independent FMAs; L[i][k] is a broadcast LDS read), so the dependent path is 16 x (scale, one FMA) -- the row-by-row
        // form waited for a k-term chain of FMAs fed by two readlanes each, 6.3 of the leaf's 86 thousand cycles.  Every sum
        // receives its terms in the same order (k ascending): same bits.
        const int c = lane & 15;
        const double *Tw = &sT[tix(wave, wave)];
        double sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = sv[k] * srd[16 * wave + k];
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i > k) sv[i] = fma(-Tw[el(i, k)], x[k], sv[i]);
        }
        if (lane < 16) {
            double *Tm = &sT[tix(wave, wave)];
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i > c) Tm[el(c, i)] = x[i];      // Dinv[i][c] at [c][i]
        }
    }
    __syncthreads();
    FVGP_STAMP();

    if (g.tiles_only) {
        // the chain's TRSM substitutes tile column by tile column (trsm_tiles_kernel) and needs these only; the full
        // 128 x 128 inverses come from one batched launch after the factorisation (launch_leaf_inverse_batched)
        const double *Tw = &sT[tix(wave, wave)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = 4 * u + q;
            double d = 0.0;
            if (a > r) d = Tw[el(r, a)];
            else if (a == r) d = srd[16 * wave + a];
            linv[wave * 256 + a * 16 + r] = d;
        }
        FVGP_STAMP();
        if (yslot && tid == 0) atomicAdd(yslot, -1);
        return;
    }
    // ---- block column `wave` of inv(L), kept in registers in MFMA B-operand layout -----------------------
    {
        const int j = wave;
        double4_t xb[8];
        // X_jj[4s+q][r]: lane (q, r) holds column r of inv(L_jj) in x[]; pick rows 4s+q
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double v0 = x[4 * s], v1 = x[4 * s + 1], v2 = x[4 * s + 2], v3 = x[4 * s + 3];
            xb[0][s] = q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : v3));
        }
#pragma unroll
        for (int m = 1; m < 8; ++m) {
            const int i = j + m;
            if (i < 8) {
                // two accumulators per product: consecutive MFMAs never wait for each other's result
                const double *Dii = &sT[tix(i, i)];
                double dd[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int cc = 4 * s + q;                       // inv(L_ii)[r][cc]
                    double d = 0.0;
                    if (r > cc) d = Dii[el(cc, r)];
                    else if (r == cc) d = srd[16 * i + r];
                    dd[s] = -d;
                }
                double4_t ta = {0.0, 0.0, 0.0, 0.0}, tb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < m; ++kk) {
                    const double *Lik = &sT[tix(i, j + kk)];
                    ta = mfma(Lik[el(r, q)], xb[kk][0], ta);
                    tb = mfma(Lik[el(r, 4 + q)], xb[kk][1], tb);
                    ta = mfma(Lik[el(r, 8 + q)], xb[kk][2], ta);
                    tb = mfma(Lik[el(r, 12 + q)], xb[kk][3], tb);
                }
                const double4_t t4 = ta + tb;
                double4_t xa = {0.0, 0.0, 0.0, 0.0}, xc = {0.0, 0.0, 0.0, 0.0};
                xa = mfma(dd[0], t4[0], xa);
                xc = mfma(dd[1], t4[1], xc);
                xa = mfma(dd[2], t4[2], xa);
                xc = mfma(dd[3], t4[3], xc);
                xb[m] = xa + xc;
            } else {
                xb[m] = (double4_t){0.0, 0.0, 0.0, 0.0};
            }
        }
        // write block column j: zero tiles above the diagonal, X_jj, then X_ij
        for (int i = 0; i < j; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) linv[(long)(16 * i + q + 4 * v) * 128 + 16 * j + r] = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = j + m;
            if (i < 8) {
#pragma unroll
                for (int v = 0; v < 4; ++v) linv[(long)(16 * i + q + 4 * v) * 128 + 16 * j + r] = xb[m][v];
            }
        }
    }
    FVGP_STAMP();
    if (yslot && tid == 0) atomicAdd(yslot, -1);
#undef FVGP_STAMP
}

}  // namespace

int launch_leaf(fvgp_handle *h, double *A, int64_t lda, double *linv, double *logdet_part, int info_base, int do_factor, int nvalid,
                int tiles_only) {
    LeafArgs g;
    g.tiles_only = tiles_only;
    g.A = A; g.lda = lda; g.linv = linv; g.logdet_part = logdet_part; g.info = h->dinfo; g.info_base = info_base;
    g.do_factor = do_factor; g.a_stride = 0; g.linv_stride = 0; g.nvalid = nvalid; g.stamps = h->leaf_stamps;
    g.yield = (h->leaf_yield && do_factor) ? h->cu_yield : nullptr;
    hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(512), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_leaf_inverse_batched(fvgp_handle *h, const double *L, int64_t ldl, int64_t nblk, double *linv) {
    if (nblk <= 0) return 0;
    LeafArgs g;
    g.A = const_cast<double *>(L); g.lda = ldl; g.linv = linv; g.logdet_part = nullptr; g.info = h->dinfo; g.info_base = 0;
    g.do_factor = 0; g.a_stride = 128 * ldl + 128; g.linv_stride = LEAF_DOUBLES; g.nvalid = 128; g.stamps = nullptr; g.tiles_only = 0; g.yield = nullptr;
    hipLaunchKernelGGL(leaf_kernel, dim3((unsigned)nblk), dim3(512), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}
