// Device code of the 128 x 128 diagonal-block Cholesky (leaf.hip: one launch per block; chain.hip: the runner of the resident
// panel kernel calls it once per 128 columns).  See leaf.hip for the algorithm and the resource shape.
#pragma once
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
    int preloaded;                // chain.hip: the block already sits in the packed LDS tiles (the load phase is skipped)
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


__device__ __forceinline__ double *uniform_ptr_rw(double *p) { return const_cast<double *>(uniform_ptr(p)); }
// 16 bytes through a buffer descriptor with sc1 (served past this compute unit's L1): data another workgroup of the SAME launch wrote
__device__ __forceinline__ double2_t ld_b128_sc1(__amdgpu_buffer_rsrc_t src, int byte_off) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src, byte_off, 0, 16);
    double2_t d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
// the inverse block: a plain store, or (CHAIN) a write-through store that the other workgroups of the launch read with sc1 loads
template <bool CHAIN>
__device__ __forceinline__ void st_linv(double *p, double v) {
    if constexpr (CHAIN) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// One block: A (128 x 128 at leading dimension g.lda) -> L in place, inverse(s) -> linv, 1 / L_ii -> logdet_part (may be null),
// first bad pivot -> g.info as info_base + column + 1.  sT: 36 x 256 doubles, srd: 128 doubles of LDS; 512 threads.
// The caller synchronises the workgroup before it reuses sT / srd.  `tid` = threadIdx.x (a caller that loops over blocks passes it
// through an opaque move per iteration, so that the compiler does not hoist every address of the body out of the loop).
template <bool CHAIN>
__device__ __forceinline__ void leaf_body(const LeafArgs &g, double *A, double *linv, double *logdet_part, const int info_base, double *sT, double *srd, const int tid) {

    // the leaf sits on the critical path of the panel chain and shares its SIMDs with trailing-update waves
    // (look-ahead): its instructions go first
    __builtin_amdgcn_s_setprio(3);
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    int nst = 0;
    // this compute unit is the leaf's while it runs: the co-resident trailing-update workgroup sleeps (gemm.hip, YIELD)
    int *const yslot = g.yield ? cu_yield_slot(g.yield) : nullptr;      // wave-uniform: lives in scalar registers
    if (yslot && tid == 0) atomicAdd(yslot, 1);
#define FVGP_STAMP() do { if (g.stamps && tid == 0) g.stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
    FVGP_STAMP();

    if (!(CHAIN && g.preloaded))
    // ---- load the lower triangle into packed tiles; strict upper of diagonal tiles <- 0 (kept inline: as a function of its own the
    //      same lines made the kernel spill 463 registers instead of 2) ----------
    {
    // eight loads of a thread in flight per LDS-write batch (two memory round trips instead of sixteen)
    const int c2 = (tid & 63) * 2, tj = c2 >> 4, rbase = tid >> 6;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(A)), 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        double2_t v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = rbase + 8 * (8 * hb + it);
            v[it] = (double2_t){0.0, 0.0};
            if (tj <= (row >> 4)) {
                if constexpr (CHAIN) v[it] = ld_b128_sc1(a_src, (int)(((long)row * g.lda + c2) * 8));
                else v[it] = *reinterpret_cast<const double2_t *>(A + (long)row * g.lda + c2);
            }
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
            if (bad >= 0 && lane == 0 && bad < g.nvalid) atomicCAS(g.info, 0, info_base + bad + 1);
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
                    atomicCAS(g.info, 0, info_base + 16 * (p + 1) + bad + 1);
            }
            __syncthreads();
            FVGP_STAMP();
        }
        // ---- L back to global (lower triangle only), tile by tile: a wave's store covers four 128-byte row segments ----
        if constexpr (CHAIN) {
            // write-through (sc1) stores of 16 bytes: lane l of a wave takes the column pair 2 (l & 7) of rows l >> 3 and 8 + (l >> 3)
            // of a tile (the other workgroups of the launch read L with sc1 loads; 8-byte sc1 stores are one fabric write each)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t s_src = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr_rw(A), 0, 0xffffffff, 0x00020000);
            for (int t = wave; t < NT; t += 8) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                const int tj = t - ti * (ti + 1) / 2;
                const double *T = &sT[t * TSZ];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int a = 8 * u + (lane >> 3), c = 2 * (lane & 7);
                    const double2_t pr = *reinterpret_cast<const double2_t *>(&T[el(a, c)]);
                    const int off = (int)((((long)(16 * ti + a)) * g.lda + 16 * tj + c) * 8);
                    if (ti != tj || c + 1 <= a) {
                        u32x4 raw;
                        __builtin_memcpy(&raw, &pr, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, s_src, off, 0, 16);
                    } else if (c == a) {
                        __hip_atomic_store(A + (long)(16 * ti + a) * g.lda + 16 * tj + c, pr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        } else {
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
        }
        // ---- the reciprocal diagonal, for the log-determinant: the logarithms are taken once, by one kernel over all blocks
        //      (neg_log_sum_kernel), not 128 at a time behind a barrier on the chain's critical path (1.4 thousand cycles) ------
        if (logdet_part != nullptr && tid < 128) logdet_part[tid] = tid < g.nvalid ? srd[tid] : 1.0;
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
            st_linv<CHAIN>(&linv[wave * 256 + a * 16 + r], d);
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
        if constexpr (CHAIN) {
            // write-through (sc1) stores, 16 bytes each: neighbouring lanes (columns r, r + 1) swap one value per pair of rows, the
            // even lane then stores rows v = 0, 2 and the odd lane rows v = 1, 3 (8-byte sc1 stores are one fabric write each:
            // the 128 KB of the inverse took 21 us that way instead of 8)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t l_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(linv)), 0, 0xffffffff, 0x00020000);
            const bool odd = r & 1;
            auto put = [&](const int i, const double x0, const double x1, const double x2, const double x3) {
#pragma unroll
                for (int vp = 0; vp < 2; ++vp) {
                    const double own = odd ? (vp ? x3 : x1) : (vp ? x2 : x0);       // the entry of the row this lane stores
                    const double send = odd ? (vp ? x2 : x0) : (vp ? x3 : x1);      // the entry of the row the neighbour stores
                    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(send), 0xB1, 0xF, 0xF, true);
                    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(send), 0xB1, 0xF, 0xF, true);
                    const double got = __hiloint2double(hi, lo);
                    const double2_t pr = odd ? (double2_t){got, own} : (double2_t){own, got};
                    const int row = 16 * i + q + 4 * (2 * vp + (odd ? 1 : 0));
                    u32x4 raw;
                    __builtin_memcpy(&raw, &pr, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, l_src, (row * 128 + 16 * j + (r & ~1)) * 8, 0, 16);
                }
            };
            for (int i = 0; i < j; ++i) put(i, 0.0, 0.0, 0.0, 0.0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = j + m;
                if (i < 8) put(i, xb[m][0], xb[m][1], xb[m][2], xb[m][3]);
            }
        } else {
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
    }
    FVGP_STAMP();
    if (yslot && tid == 0) atomicAdd(yslot, -1);
#undef FVGP_STAMP
}

}  // namespace
