// Device code of the 128 x 128 diagonal-block Cholesky (leaf.hip: one launch per block; chain.hip: the runner of the resident
// panel kernel calls it once per 128 columns).  See leaf.hip for the algorithm and the resource shape.
#pragma once
#include "common.h"
#include <type_traits>

namespace {


constexpr int TSZ = 256;          // a packed 16x16 tile, unpadded: 36 tiles = exactly 72 KB
constexpr int NT = 36;            // tiles (i,j), j <= i < 8
constexpr int LEAF_SRD = 128 + 2;      // doubles behind the tiles: 1 / L_aa; the solvers' counter of the factor loop

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
    unsigned long long *col_flag; // chain.hip: raised to col_base + c + 1 once the 16-column tile column c of L and the inverse of its diagonal
    unsigned long long col_base;  //            tile are in memory (the block rows of the panel's square solve BEHIND the running leaf); may be null
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

// diagnostic build only (tools/build_variant.sh fine -DFVGP_LEAF_FINE; tools/leaf_fine.py): cycle stamps inside PanelBlock
#ifdef FVGP_LEAF_FINE
__device__ unsigned long g_fine[512];
__device__ int g_fine_n;
#define FVGP_FINE() do { } while (0)
__device__ unsigned long g_wfine[8 * 8 * 8];          // [wave][step][event]: s_memtime of every wave's events in the factor loop
#define FVGP_WFINE(p, k) do { if (lane == 0) g_wfine[(wave * 8 + (p)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FVGP_FINE() do { } while (0)
#define FVGP_WFINE(p, k) do { } while (0)
#endif

// Four-column block B of a 16x16 tile.  The tile is kept as the full symmetric matrix in the MFMA accumulator layout, lane
// (r, q) holds D[q + 4v][r] = D[r][q + 4v], so accumulator slot B of lane (r, q) IS the panel entry (row r, panel column q).
//   1. the 4x4 diagonal block is factored as wave-uniform scalars (its ten entries by readlane): the pivot chain -- four
//      reciprocal square roots in sequence -- waits for nothing else;
//   2. its inverse W (4x4, lower) follows from the same scalars, column q in lane group q (a dozen flops), and lane (r < 4, q)
//      keeps W[r][q]: that register IS the A operand of ONE MFMA that solves every row of the panel against the block,
//      X^T = W R^T (B operand: the accumulator slot as it stands), the result landing in slot 0 of the same lanes.  W's six
//      entries below its diagonal also go to LDS, transposed into the (free) strict upper half of the 4x4 diagonal block they
//      invert; its diagonal is 1 / L_rr (srd): the other waves solve their 16 x 16 tiles of the rows below with the same MFMA
//      (tile_trsm) instead of a per-row substitution, and the inverse of the whole tile follows the same way at the end;
//   3. the factored panel is both operands of ONE rank-4 MFMA update of the rest of the tile.
// (Versions before: 16 columns swept with 15 - J broadcast-and-FMA updates behind each pivot, 7.4k cycles per tile; the panel
// rows solved per lane group with three dependent cross-lane broadcasts, 4.6k.)
// per-lane constants of PanelBlock: dq[k] = (q == k), mr[k] = (r == k) as doubles (selects by multiplication: one of four terms survives)
struct PanelLane { double dq[4], mr[4]; };
__device__ __forceinline__ PanelLane panel_lane(const int r, const int q) {
    PanelLane c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c.dq[k] = q == k ? 1.0 : 0.0; c.mr[k] = r == k ? 1.0 : 0.0; }
    return c;
}

template <int B>
struct PanelBlock {
    static __device__ __forceinline__ void step(double4_t &acc, double *sT, double *srdt, const PanelLane &pc, int r, int q, int lane,
                                                double (&wops)[4], double (&lcs)[4]) {
        constexpr int R0 = 4 * B;
        FVGP_FINE();
        const bool above = r < R0 + q;                        // finished rows, and the 4x4 block above its diagonal
        const double x = above ? 0.0 : acc[B];
        const double d00 = bcast<R0>(x), d10 = bcast<R0 + 1>(x), d20 = bcast<R0 + 2>(x), d30 = bcast<R0 + 3>(x);
        const double d11 = bcast<16 + R0 + 1>(x), d21 = bcast<16 + R0 + 2>(x), d31 = bcast<16 + R0 + 3>(x);
        const double d22 = bcast<32 + R0 + 2>(x), d32 = bcast<32 + R0 + 3>(x);
        const double d33 = bcast<48 + R0 + 3>(x);
        FVGP_FINE();
        // (no test of the pivots here: a pivot <= 0 or NaN makes its reciprocal square root, and everything after it, inf or NaN;
        // diag_factor_acc looks at the sixteen 1 / L_rr once per tile and finds the first that is not a positive finite number)
        const double y0 = rsqrt_nr(d00);
        const double l10 = d10 * y0, l20 = d20 * y0, l30 = d30 * y0;
        const double e11 = fma(-l10, l10, d11);
        const double y1 = rsqrt_nr(e11);
        const double l21 = fma(-l20, l10, d21) * y1, l31 = fma(-l30, l10, d31) * y1;
        const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
        const double y2 = rsqrt_nr(e22);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * y2;
        const double e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        const double y3 = rsqrt_nr(e33);
        FVGP_FINE();
        // column q of W = inv(L_bb) by forward substitution on the unit vector e_q, in lane group q
        const double w0 = pc.dq[0] * y0;
        const double w1 = fma(-l10, w0, pc.dq[1]) * y1;
        const double w2 = fma(-l21, w1, fma(-l20, w0, pc.dq[2])) * y2;
        const double w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, pc.dq[3]))) * y3;
        const double wop = fma(pc.mr[3], w3, fma(pc.mr[2], w2, fma(pc.mr[1], w1, pc.mr[0] * w0)));      // W[r][q], rows >= 4 of the operand: 0
        if (r < 4 && q < r) sT[el(R0 + q, R0 + r)] = wop;
        if (lane == 0) { srdt[R0] = y0; srdt[R0 + 1] = y1; srdt[R0 + 2] = y2; srdt[R0 + 3] = y3; }      // 1 / L_rr
        FVGP_FINE();
        // every row of the panel: X[r][R0 + q] = sum_k W[q][k] R[r][R0 + k], one MFMA, result in slot 0
        double4_t xs = {0.0, 0.0, 0.0, 0.0};
        xs = mfma(wop, x, xs);
        double xf = xs[0];
        FVGP_FINE();
        // the diagonal entries with one correction step (on this lane group's pivot), zeros above them
        const double eq = fma(pc.dq[3], e33, fma(pc.dq[2], e22, fma(pc.dq[1], e11, pc.dq[0] * d00)));
        const double yq = fma(pc.dq[3], y3, fma(pc.dq[2], y2, fma(pc.dq[1], y1, pc.dq[0] * y0)));
        const double pd = sqrt_from(eq, yq);
        if (r == R0 + q) xf = pd;
        if (above) xf = 0.0;
        if (r >= R0 + q) sT[el(r, R0 + q)] = xf;
        wops[B] = wop; lcs[B] = xf;                              // what tile_trsm reads back from LDS, for the wave that has them anyway
        FVGP_FINE();
        if constexpr (B < 3) {
            acc = mfma(-xf, xf, acc);
            PanelBlock<B + 1>::step(acc, sT, srdt, pc, r, q, lane, wops, lcs);
        }
    }
};

// the symmetric 16 x 16 tile at sT (lower part valid) in the accumulator layout diag_factor works on
__device__ __forceinline__ double4_t load_sym_tile(const double *sT, int r, int q) {
    double4_t acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int i = q + 4 * v; acc[v] = (i >= r) ? sT[el(i, r)] : sT[el(r, i)]; }
    return acc;
}

// one wave: Cholesky of the symmetric 16x16 tile in `acc` into sT (lower part; the inverses of its 4x4 diagonal blocks into their
// strict upper halves); 1/diag -> srd[0..15].  Returns the first bad pivot column (not positive, or NaN: dpotrf's info) or -1
__device__ __forceinline__ int diag_factor_acc(double4_t acc, double *sT, double *srd, const PanelLane &pc, int lane, double (&wops)[4], double (&lcs)[4]) {
    const int r = lane & 15, q = lane >> 4;
    PanelBlock<0>::step(acc, sT, srd, pc, r, q, lane, wops, lcs);
    // a pivot that is not positive leaves inf or NaN in its 1 / L_rr and NaN in every later one: the first such column, if any
    // (the four uniform values of the last block are in registers; the twelve before them are read back: off the critical path)
    const double yr = srd[r];
    const unsigned long long m = __ballot(!(yr > 0.0 && yr < 1.0e300)) & 0xffffull;
    return m ? __builtin_ctzll(m) : -1;
}

// the W operand of block b of the factored diagonal tile Lp (PanelBlock, step 2) as lane (r, q) supplies it: W[r][q] for r < 4.
// Two unconditional LDS reads (a lane with nothing to fetch reads the block's first entries) and two selects: no divergent branch.
__device__ __forceinline__ double w_operand(const double *Lp, const double *srdp, const int b, const int r, const int q) {
    const bool lower = r < 4 && q < r, diag = r < 4 && q == r;
    const double v = Lp[lower ? el(4 * b + q, 4 * b + r) : el(4 * b, 4 * b)];
    const double d = srdp[4 * b + (r & 3)];
    return lower ? v : (diag ? d : 0.0);
}

// one wave: the 16 x 16 tile of the rows below the diagonal tile L_pp <- tile inv(L_pp)^T, four columns at a time: the MFMA of
// PanelBlock step 2 with the inverted 4x4 diagonal blocks wave 0 left beside L_pp (Lp: the factored diagonal tile, srdp: its
// 1 / L_rr), then a rank-4 update of the remaining columns with L_pp's own columns.  The tile is kept transposed in the
// accumulator layout: on entry lane (a, q) slot v = T[a][q + 4v]; on return xres[b] = X[a][4b + q].  (Lp's entries above its
// diagonal only ever meet accumulator rows that are finished.)
__device__ __forceinline__ void tile_trsm_acc(double4_t acc, const double *Lp, const double *srdp, const int r, const int q, double (&xres)[4]) {
    double wop[4], lc[3];
#pragma unroll
    for (int b = 0; b < 4; ++b) wop[b] = w_operand(Lp, srdp, b, r, q);
#pragma unroll
    for (int b = 0; b < 3; ++b) lc[b] = Lp[el(r, 4 * b + q)];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double4_t xs = {0.0, 0.0, 0.0, 0.0};
        xs = mfma(wop[b], acc[b], xs);
        xres[b] = xs[0];
        if (b < 3) acc = mfma(-lc[b], xres[b], acc);
    }
}
// the same with the operands in registers (wave 0: PanelBlock has just produced them)
__device__ __forceinline__ void tile_trsm_regs(double4_t acc, const double (&wop)[4], const double (&lc)[4], double (&xres)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double4_t xs = {0.0, 0.0, 0.0, 0.0};
        xs = mfma(wop[b], acc[b], xs);
        xres[b] = xs[0];
        if (b < 3) acc = mfma(-lc[b], xres[b], acc);
    }
}
__device__ __forceinline__ void tile_trsm(double *Ti, const double *Lp, const double *srdp, const int r, const int q, double (&xres)[4]) {
    double4_t acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = Ti[el(r, q + 4 * v)];
    tile_trsm_acc(acc, Lp, srdp, r, q, xres);
#pragma unroll
    for (int b = 0; b < 4; ++b) Ti[el(r, q + 4 * b)] = xres[b];
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
// VERIFY (chain.hip, option "chain_verify"): *vsum receives this thread's share of the sum of the bit patterns of what the block hands
// to the other workgroups of the launch -- L's strictly lower 16 x 16 tiles and the inverses of its diagonal tiles, as stored
template <bool CHAIN, bool VERIFY = false>
__device__ __forceinline__ void leaf_body(const LeafArgs &g, double *A, double *linv, double *logdet_part, const int info_base, double *sT, double *srd, const int tid,
                                          unsigned long long *vsum = nullptr) {

    // the leaf sits on the critical path of the panel chain and shares its SIMDs with trailing-update waves
    // (look-ahead): its instructions go first
    __builtin_amdgcn_s_setprio(3);
    int nst = 0;
    // this compute unit is the leaf's while it runs: the co-resident trailing-update workgroup sleeps (gemm.hip, YIELD)
    int *const yslot = g.yield ? cu_yield_slot(g.yield) : nullptr;      // wave-uniform: lives in scalar registers
    if (yslot && tid == 0) atomicAdd(yslot, 1);
#define FVGP_STAMP() do { if (g.stamps && tid == 0) g.stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
    FVGP_STAMP();

    if (!(CHAIN && g.preloaded))
    // ---- load the lower triangle into the packed tiles by LDS-DMA: a wave instruction lands 1 KB = rows 8 u .. 8 u + 7 of one tile
    //      (lane l at 16 l bytes: row 8 u + (l >> 3), position l & 7 of the row's eight 16-byte pieces, which holds piece
    //      (l & 7) ^ ((row >> 1) & 7) of the tile's row: the tiles' own column swizzle, el()); 72 half-tiles, nine per wave, all in
    //      flight: ONE memory round trip and no staging registers (as register loads in two batches the phase took 6.3 thousand
    //      cycles).  The strict upper halves of the diagonal tiles -- whatever the matrix holds there -- are then zeroed.
    {
        typedef __attribute__((address_space(3))) void lds_void;
        const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
        const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(A)), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int hx = wave_u + 8 * it;                      // half-tile 0 .. 71: lower tile hx >> 1 (row-major over the triangle), rows 8 (hx & 1) ..
            const int pt = hx >> 1, u = hx & 1;
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= pt) ++ti;
            const int tj = pt - ti * (ti + 1) / 2;
            const int a = 8 * u + (ln >> 3);
            const int piece = (ln & 7) ^ ((a >> 1) & 7);
            const int voff = (int)(((long)(16 * ti + a) * g.lda + 16 * tj + 2 * piece) * 8);
            lds_void *dst = (lds_void *)&sT[pt * TSZ + 8 * u * 16];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_src, dst, 16, voff, 0, 0, CHAIN ? 16 : 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double *Td = &sT[tix(wave_u, wave_u)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * ln + k, a = e >> 4, c = e & 15;
            if (c > a) Td[el(a, c)] = 0.0;
        }
    }
    if (tid == 0) *reinterpret_cast<int *>(srd + 128) = 0;      // the solvers' counter of the factor loop
    __syncthreads();
    FVGP_STAMP();
    // the lane coordinates of everything below come from an opaque copy of the thread index: derived from `tid` itself the compiler
    // hoists the factor loop's per-lane addresses above the load phase, and the registers they hold there push its sixteen loads in
    // flight out to scratch (458 spilled registers)
    int tid_f = tid;
    asm volatile("" : "+v"(tid_f));
    const int lane = tid_f & 63, wave = __builtin_amdgcn_readfirstlane(tid_f >> 6);
    const int r = lane & 15, q = lane >> 4;

    // one finished 16 x 16 tile (ti, tj) of L from LDS to global memory (lower triangle of the block only)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t s_src = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr_rw(A), 0, 0xffffffff, 0x00020000);
    auto store_tile = [&](const int ti, const int tj) {
        const double *T = &sT[tix(ti, tj)];
        if constexpr (CHAIN) {
            // write-through (sc1) stores of 16 bytes: lane l of a wave takes the column pair 2 (l & 7) of rows l >> 3 and 8 + (l >> 3)
            // of a tile (the other workgroups of the launch read L with sc1 loads; 8-byte sc1 stores are one fabric write each)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = 4 * u + q;
                if (ti != tj || r <= a) A[(long)(16 * ti + a) * g.lda + 16 * tj + r] = T[el(a, r)];
            }
        }
    };

    // one wave: inv(L_tt)^T = I inv(L_tt)^T is the tile solve of the factor loop applied to the identity; the strictly lower part of the
    // inverse goes, transposed, into the tile's strict upper half (Dinv[i][c] at [c][i]; the diagonal is srd), and -- `tiles_only`, what the
    // panel chain asks for -- the inverse itself to linv (8 x 256 doubles, zeros above the diagonals)
    [[maybe_unused]] unsigned long long vs_inv = 0ull;      // VERIFY: bit patterns of the inverse entries this thread stored
    auto tile_inverse = [&](const int t) {
        double4_t id;
#pragma unroll
        for (int v = 0; v < 4; ++v) id[v] = (r == q + 4 * v) ? 1.0 : 0.0;
        double xinv[4];
        double *Tt = &sT[tix(t, t)];
        tile_trsm_acc(id, Tt, &srd[16 * t], r, q, xinv);            // xinv[b] = inv(L_tt)[4b + q][r]: exact zeros above the diagonal, srd on it
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int a = 4 * b + q;
            if (a > r) Tt[el(r, a)] = xinv[b];
            if constexpr (!CHAIN) { if (g.tiles_only) linv[t * 256 + a * 16 + r] = xinv[b]; }      // (CHAIN: publish_column, below)
        }
    };
    // CHAIN: wave 4 -- idle in the factor loop -- sends a FINISHED tile column c to memory, a step behind the waves that produce it: the
    // tiles (c, c) .. (7, c) of L and the inverse of the diagonal tile as tile_inverse left it in LDS (below the diagonal transposed in
    // the tile's strict upper half, the diagonal in srd), all write-through; when they have left it raises the column flag.  One wave:
    // its own s_waitcnt covers every store, no barrier.  The workgroups that solve against this block (chain.hip, follow) start on
    // column c while the factorisation is at column c + 2.
    [[maybe_unused]] auto publish_column = [&](const int c, const bool raise) {
        if constexpr (CHAIN) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            for (int i = c; i < 8; ++i) store_tile(i, c);
            const double *Tt = &sT[tix(c, c)];
            const __amdgpu_buffer_rsrc_t l_src = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr_rw(linv), 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int a = 8 * u + (lane >> 3), cc = 2 * (lane & 7);
                const double up0 = Tt[el(cc, a)], up1 = Tt[el(cc + 1, a)], dg = srd[16 * c + a];
                const double2_t pr = {a > cc ? up0 : (a == cc ? dg : 0.0), a > cc + 1 ? up1 : (a == cc + 1 ? dg : 0.0)};
                if constexpr (VERIFY) vs_inv += (unsigned long long)__double_as_longlong(pr[0]) + (unsigned long long)__double_as_longlong(pr[1]);
                u32x4 raw;
                __builtin_memcpy(&raw, &pr, 16);
                __builtin_amdgcn_raw_buffer_store_b128(raw, l_src, (c * 256 + a * 16 + cc) * 8, 0, 16);
            }
            if (raise && g.col_flag) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(g.col_flag, g.col_base + (unsigned long long)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    double4_t xb0;          // inv(L_ww)[4s + q][r] of this wave's diagonal tile (the seed of the block-column inverse below)
    if (g.do_factor) {
        // Schedule of the eight 16-column steps.  Wave 0 owns the chain of diagonal tiles and never waits inside a step: behind the
        // step's barrier it solves tile (p+1, p) against L_pp (tile_trsm), applies the solved rows -- still in its registers, the
        // accumulator layout is both MFMA operands of X X^T -- to the symmetric tile (p+1, p+1) and factors it on the spot.  Waves
        // 1, 2, 3, 5, 6, 7 solve the other tiles of column p, meet at a counter in LDS (wave 0 adds to it once its tile is
        // stored), send the finished column p to global memory and update the tiles to the right.  Wave 4 shares wave 0's SIMD --
        // fp64 MFMA and the vector ALU share a pipe -- and takes no work during the steps.  ONE workgroup barrier per step.
        int *const s_cnt = reinterpret_cast<int *>(srd + 128);         // (zeroed before the barrier behind the load phase)
        const int slot = (wave == 0 || wave == 4) ? -1 : (wave < 4 ? wave - 1 : wave - 2);
        double wops[4], lcs[4];                                        // wave 0: the solve's operands of the diagonal tile it has just factored
        const PanelLane pc = panel_lane(r, q);
        if (wave == 0) {
            const int bad = diag_factor_acc(load_sym_tile(&sT[tix(0, 0)], r, q), &sT[tix(0, 0)], &srd[0], pc, lane, wops, lcs);
            if (bad >= 0 && lane == 0 && bad < g.nvalid) atomicCAS(g.info, 0, info_base + bad + 1);
        }
        FVGP_STAMP();
        __syncthreads();
        FVGP_STAMP();
#pragma nounroll
        for (int p = 0; p < 7; ++p) {
            const double *Lp = &sT[tix(p, p)];
            const double *srdp = &srd[16 * p];
            FVGP_WFINE(p, 0);
            if (wave == 0) {
                double *Td = &sT[tix(p + 1, p + 1)], *T1 = &sT[tix(p + 1, p)];
                double4_t xacc;
#pragma unroll
                for (int v = 0; v < 4; ++v) xacc[v] = T1[el(r, q + 4 * v)];
                double4_t dacc = load_sym_tile(Td, r, q);             // (issued before the solve: its latency hides under it)
                double xres[4];
                tile_trsm_regs(xacc, wops, lcs, xres);
#pragma unroll
                for (int b = 0; b < 4; ++b) T1[el(r, q + 4 * b)] = xres[b];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                FVGP_STAMP();
                FVGP_WFINE(p, 1);
                // the solved rows are both operands of X X^T: two accumulators, so that consecutive MFMAs do not wait for each other
                double4_t dac2 = {0.0, 0.0, 0.0, 0.0};
                dacc = mfma(-xres[0], xres[0], dacc);
                dac2 = mfma(-xres[1], xres[1], dac2);
                dacc = mfma(-xres[2], xres[2], dacc);
                dac2 = mfma(-xres[3], xres[3], dac2);
                dacc += dac2;
                FVGP_STAMP();
                const int bad = diag_factor_acc(dacc, Td, &srd[16 * (p + 1)], pc, lane, wops, lcs);
                if (bad >= 0 && lane == 0 && 16 * (p + 1) + bad < g.nvalid)
                    atomicCAS(g.info, 0, info_base + 16 * (p + 1) + bad + 1);
                FVGP_STAMP();
                FVGP_WFINE(p, 5);
            } else if (slot >= 0) {
                const int it = p + 2 + slot;
                if (it < 8) { double xr[4]; tile_trsm(&sT[tix(it, p)], Lp, srdp, r, q, xr); }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                FVGP_WFINE(p, 1);
                // every solved tile of column p is in LDS once the seven solvers of this step have added to the counter
                while (__hip_atomic_load(s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 7 * (p + 1)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                FVGP_WFINE(p, 2);
                // ---- LEFT-looking update of what the next step needs: the tiles (i, p+1), i >= p+2, and the diagonal tile (p+2, p+2)
                //      receive ALL their terms C -= sum_{k <= p} X_ik X_jk^T now (the diagonal tile its last one, k = p+1, from wave 0's
                //      registers in the next step).  A right-looking update does the same flops as 27, 20, 14, .. tiles in the first
                //      steps, where these six waves are what wave 0 then waits for; this way a step has at most 16 tile products. ----
                const int ntarget = p <= 5 ? 7 - p : 0;                    // (p = 6: column 7 is the diagonal tile alone, wave 0's)
                for (int t = slot; t < ntarget; t += 6) {
                    const int i = t < 6 - p ? p + 2 + t : p + 2, j = t < 6 - p ? p + 1 : p + 2;
                    double *C = &sT[tix(i, j)];
                    const double *Xi = &sT[tix(i, 0)], *Xj = &sT[tix(j, 0)];      // tile (i, k) = Xi + k TSZ
                    const int o0 = el(r, q), o1 = el(r, 4 + q), o2 = el(r, 8 + q), o3 = el(r, 12 + q);
                    double4_t acc, ac2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] = C[el(q + 4 * v, r)];
                    double a0 = Xi[o0], a1 = Xi[o1], a2 = Xi[o2], a3 = Xi[o3], b0 = Xj[o0], b1 = Xj[o1], b2 = Xj[o2], b3 = Xj[o3];
                    for (int k = 0; k <= p; ++k) {
                        const int kn = k < p ? (k + 1) * TSZ : k * TSZ;
                        const double n0 = Xi[kn + o0], n1 = Xi[kn + o1], n2 = Xi[kn + o2], n3 = Xi[kn + o3];
                        const double m0 = Xj[kn + o0], m1 = Xj[kn + o1], m2 = Xj[kn + o2], m3 = Xj[kn + o3];
                        acc = mfma(-a0, b0, acc);
                        ac2 = mfma(-a1, b1, ac2);
                        acc = mfma(-a2, b2, acc);
                        ac2 = mfma(-a3, b3, ac2);
                        a0 = n0; a1 = n1; a2 = n2; a3 = n3; b0 = m0; b1 = m1; b2 = m2; b3 = m3;
                    }
                    acc += ac2;
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (i != j || r <= q + 4 * v) C[el(q + 4 * v, r)] = acc[v];
                }
                FVGP_WFINE(p, 3);
                // ---- column p is final: its 8 - p tiles go to global memory while wave 0 factors the next diagonal tile ----
                if constexpr (!CHAIN) for (int i = p + slot; i < 8; i += 6) store_tile(i, p);      // (CHAIN: wave 4, a step later)
                FVGP_WFINE(p, 4);
                // ---- and so is the diagonal tile (p, p): its inverse (transposed into its strict upper half, over the 4x4 block
                //      inverses nobody reads any more; the diagonal is srd) by the wave with the least to do in a step ----
                if (slot == 5) tile_inverse(p);
                FVGP_WFINE(p, 5);
            } else if (CHAIN && p > 0) {
                publish_column(p - 1, true);
            }
            __syncthreads();
            FVGP_WFINE(p, 6);
            FVGP_STAMP();
        }
        if constexpr (CHAIN) { if (wave == 4) publish_column(6, true); }
        else if (wave == 1) store_tile(7, 7);
        // ---- the reciprocal diagonal, for the log-determinant: the logarithms are taken once, by one kernel over all blocks
        //      (neg_log_sum_kernel), not 128 at a time behind a barrier on the chain's critical path (1.4 thousand cycles) ------
        if (logdet_part != nullptr && tid < 128) logdet_part[tid] = tid < g.nvalid ? srd[tid] : 1.0;
        FVGP_STAMP();
        if (wave == 7) tile_inverse(7);
        __syncthreads();
        if constexpr (CHAIN) { if (wave == 4) publish_column(7, false); }      // (the caller publishes the whole block behind this)
        // the seed of the block-column inverse below (not needed when only the tile inverses go out)
        if (!g.tiles_only) {
            const double *Tw = &sT[tix(wave, wave)];
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int a = 4 * b + q; xb0[b] = a > r ? Tw[el(r, a)] : (a == r ? srd[16 * wave + a] : 0.0); }
        }
        FVGP_STAMP();
    } else {
        if (tid < 128) srd[tid] = 1.0 / sT[tix(tid >> 4, tid >> 4) + el(tid & 15, tid & 15)];
        __syncthreads();
        // ---- inverse of the 8 diagonal tiles of a given factor: wave w inverts tile (w,w); its strictly-lower part goes,
        //      transposed, into the tile's (unused) strict upper half, the diagonal is srd -------------------
        // lane c solves L x = e_c (column c of the inverse) by a COLUMN sweep: a finished x[k] goes into every later row at once
        // (independent FMAs; L[i][k] is a broadcast LDS read), so the dependent path is 16 x (scale, one FMA)
        const int c = lane & 15;
        const double *Tw = &sT[tix(wave, wave)];
        double x[16], sv[16];
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
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const double v0 = x[4 * s4], v1 = x[4 * s4 + 1], v2 = x[4 * s4 + 2], v3 = x[4 * s4 + 3];
            xb0[s4] = q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : v3));
        }
        __syncthreads();
        FVGP_STAMP();
    }

    if (g.tiles_only) {
        // the chain's TRSM substitutes tile column by tile column (trsm_tiles_kernel / trsm_sub) and needs the tile inverses only; the
        // full 128 x 128 inverses come from one batched launch after the factorisation (launch_leaf_inverse_batched)
        if (!g.do_factor) {           // (tile inverses of a given factor: as left above, transposed in the strict upper halves)
            const double *Tw = &sT[tix(wave, wave)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = 4 * u + q;
                double d = 0.0;
                if (a > r) d = Tw[el(r, a)];
                else if (a == r) d = srd[16 * wave + a];
                st_linv<CHAIN>(&linv[wave * 256 + a * 16 + r], d);
            }
        }
        if constexpr (VERIFY) {           // + the 28 strictly lower tiles: 7168 doubles, fourteen per thread
            unsigned long long vs = vs_inv;
            for (int e = tid; e < 28 * TSZ; e += 512) {
                const int p = e / TSZ;
                int ti = 1;
                while (ti * (ti + 1) / 2 <= p) ++ti;
                const int tj = p - ti * (ti - 1) / 2;
                vs += (unsigned long long)__double_as_longlong(sT[tix(ti, tj) + (e - p * TSZ)]);
            }
            *vsum = vs;
        }
        FVGP_STAMP();
        if (yslot && tid == 0) atomicAdd(yslot, -1);
        return;
    }
    // ---- block column `wave` of inv(L), kept in registers in MFMA B-operand layout -----------------------
    {
        const int j = wave;
        double4_t xb[8];
        xb[0] = xb0;        // X_jj[4s+q][r]
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
