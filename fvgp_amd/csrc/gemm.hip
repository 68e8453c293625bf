// fp64 MFMA GEMM / SYRK for gfx950:  C = alpha * opA * opB + beta * C  on 128x128 tiles.
// This is the kernel the metric lives in: the trailing update of the blocked Cholesky (dsyrk/dgemm inside LAPACK dpotrf, reached
// from fvgp/gp_lin_alg.py:245) is A22 -= L21 * L21^T (a_kmajor = 0, b_nmajor = 0, lower = 1, ROLE 1); every other level-3 step
// of the path (panel products, POTRI, posterior cross products) is one of the four operand layouts below.
// Design (CDNA4):
//   * v_mfma_f64_16x16x4_f64; one 256-thread workgroup = 4 waves in a 2x2 grid, each wave owns 64x64 of C = 16 MFMA tiles =
//     64 fp64 accumulators per lane; 2 workgroups per CU (<= 228 VGPR, 76 KB LDS each), so one workgroup's C read-modify-write
//     epilogue hides under the other's MFMA stream;
//   * K is walked in steps of 16 through a double-buffered LDS image, one barrier per step;
//   * both operands k-minor ((M,K) x (N,K): the trailing update, every panel product, POTRI, the posterior substitution):
//     unpadded [128][16] images whose 16-byte chunks are XOR-swizzled per row (bank-conflict free for ds_read_b128's lane
//     groups), filled by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write), lane group q owns k = 4q .. 4q+3 of a
//     step (one 16-byte read per fragment pair), all sixteen reads of a step issued before its first MFMA, every address
//     loop-invariant (buffer descriptors with a constant lane offset and the K position in an SGPR, the loop unrolled over the
//     two LDS buffers): no vector-ALU instruction in the loop (fp64 MFMA and the VALU share a pipe);
//   * the other layouts ((K,M) and / or (K,N) operands) keep a plain loop: global -> registers -> LDS, padded images
//     (k-minor [128][16+2], m-minor [16][128+16]), 8-byte fragment reads;
//   * ROLE 1 (the trailing update) polls its compute unit's yield counter once per K step with a scalar load and sleeps while a
//     workgroup of the panel chain runs there (common.h, cu_yield_slot);
//   * blockIdx -> tile map: 8x8 super-tiles dealt to the XCDs (blocks b, b+8, .. share an L2), either by formula or from an
//     XCD-balanced tile table built on the host per launch shape.
#include "common.h"
#include <cstring>
#include <type_traits>

namespace {

constexpr int BK = 16;
constexpr int LDK = 18;            // doubles per row of a padded k-minor LDS image  [128][18]
constexpr int LDM = 144;           // doubles per row of an m-minor LDS image [16][144]
constexpr int IMG = 128 * LDK;     // == 16 * LDM == 2304 doubles per operand image

struct GemmArgs {
    const double *A; const double *B; double *C;
    long lda, ldb, ldc;
    double alpha, beta;
    int tiles_m, tiles_n, lower;      // lower: 0 all tiles, 1 tj <= ti, 2 tj <= ti*ls + lo (row-sharded trailing update)
    int ls, lo;
    int bcr, bcb, bco;                // B rows in all-gather (block-cyclic) order, see GemmDesc
    int rev;                          // tile rows enumerated last to first
    long K;
    long kb0, kbi, kbj, ke0, kei, kej;
    long ntiles;
    const int *tab;                   // balanced block -> tile table ((ti << 16) | tj, -1 = no tile), or nullptr: formula
    long ksplit, csplit;              // split-K (gridDim.y > 1): block row y takes K elements [y ksplit, (y+1) ksplit), its tile goes to C + y csplit
    int tri;                          // split-K against a lower-triangular B (N, K): tile column tj stops at K = 128 (tj + 1), slices past that are nobody's
    int ny;                           // strided batch (ksplit == 0, gridDim.y > 1): problem (y, z) = (blockIdx.y % ny, blockIdx.y / ny)
    long ab1, ab2, bb1, bb2, cb1, cb2;   // takes its operands at A + y ab1 + z ab2, B + .., C + ..
    const int *yield;                 // trailing update (ROLE 1): per-CU counters raised by a co-resident workgroup of the panel chain
    int *raise;                       // chain kernels (small tiles, K = 128): raise this CU's counter while a workgroup runs (nullptr: no)
};

// linear index -> (ti, tj).  Tiles are enumerated in super-tiles of 8 x SN (SN = min(8, tiles_n)),
// row-major inside a super-tile; in lower mode only super-tiles that touch ti >= tj exist
// (super-row si holds min(si+1, sn) of them), so neighbours in the order share operand slabs.
__host__ __device__ inline void tile_of(long t, int tiles_m, int tiles_n, int lower, int &ti, int &tj) {
    constexpr int S = 8;
    const int SN = tiles_n < S ? tiles_n : S;
    const int sn = (tiles_n + SN - 1) / SN;
    const long per = (long)S * SN;
    const long st = t / per; const int in = (int)(t % per);
    int si, sj;
    if (!lower || SN < S) {          // SN < S implies sn == 1
        si = (int)(st / sn); sj = (int)(st % sn);
    } else {
        const long tri = (long)sn * (sn + 1) / 2;
        if (st < tri) {
            si = (int)((__builtin_sqrt(8.0 * (double)st + 1.0) - 1.0) * 0.5);
            while ((long)(si + 1) * (si + 2) / 2 <= st) ++si;
            while ((long)si * (si + 1) / 2 > st) --si;
            sj = (int)(st - (long)si * (si + 1) / 2);
        } else {
            const long rr = st - tri;
            si = sn + (int)(rr / sn); sj = (int)(rr % sn);
        }
    }
    ti = si * S + in / SN; tj = sj * SN + in % SN;
}

// row-sharded trailing update (lower == 2, tile (ti, tj) exists iff tj <= ti*ls + lo): super-row si holds only
// the super-tiles its widest row reaches, so the grid carries no dead half.  Linear search over the
// super-rows (a few dozen scalar iterations per workgroup).
__host__ __device__ inline int rs_super_count(int si, int tiles_n, int SN, int ls, int lo) {
    long w = (long)(si * 8 + 7) * ls + lo + 1;
    if (w <= 0) return 0;
    if (w > tiles_n) w = tiles_n;
    return (int)((w + SN - 1) / SN);
}
__host__ __device__ inline void tile_of_rs(long t, int tiles_m, int tiles_n, int ls, int lo, int &ti, int &tj) {
    constexpr int S = 8;
    const int SN = tiles_n < S ? tiles_n : S;
    const long per = (long)S * SN;
    long st = t / per; const int in = (int)(t % per);
    const int sm = (tiles_m + S - 1) / S;
    int si = 0;
    for (; si < sm; ++si) {
        const int c = rs_super_count(si, tiles_n, SN, ls, lo);
        if (st < c) break;
        st -= c;
    }
    ti = si * S + in / SN; tj = (int)st * SN + in % SN;      // si == sm (past the end) gives ti >= tiles_m: exits
}

__host__ __device__ inline long xcd_remap(long b, long nwg, int tiles_n) {
    const long per = 8L * (tiles_n < 8 ? tiles_n : 8);      // tiles per super-tile
    const long nst = nwg / per;                             // grid is a whole number of super-tiles
    const long xcd = b % 8, idx = b / 8;                    // idx-th block of this XCD
    const long full = nst / 8 * 8;                          // super-tiles dealt round-robin
    // round r of eight super-tiles goes to the XCDs in alternating direction (0..7, 7..0, ..): when the work per
    // super-tile falls monotonically along the enumeration (per-tile K ranges of the triangular inverse) every XCD
    // still gets the same total
    const long round = idx / per;
    const long st = round * 8 + ((round & 1) ? 7 - xcd : xcd);
    if (st < full) return st * per + idx % per;
    // leftover super-tiles (< 8): spread their tiles over all XCDs in plain order
    const long rem_blocks = nwg - full * per;
    const long k = b - (nwg - rem_blocks);                  // only reached by the last rem_blocks blocks
    return full * per + (k >= 0 ? k : 0);
}

// epilogue: lane holds D[row = q + 4v][col = r] of each 16x16 MFMA tile.  The read-modify-write of C is done in two batches of
// 32 loads per lane, all issued before the first use, so a tile pays two memory round trips instead of sixteen.
__device__ __forceinline__ void store_tile(double4_t (&acc)[4][4], double *cbase, long ldc, double alpha, double beta) {
    if (beta != 0.0) {
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            double old[2][4][4];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        old[ii][v][j] = cbase[((ih * 2 + ii) * 16 + 4 * v) * ldc + j * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        cbase[((ih * 2 + ii) * 16 + 4 * v) * ldc + j * 16] = alpha * acc[ih * 2 + ii][j][v] + beta * old[ii][v][j];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cbase[(i * 16 + 4 * v) * ldc + j * 16] = alpha * acc[i][j][v];
    }
}

// ROLE only names the instantiation (0 generic, 1 trailing update of the Cholesky) so that profilers list the kernel the metric
// lives in under its own symbol; ROLE 1 also yields its compute unit to the panel chain (see the K loop)
template <int AKM, int BNM, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    // 76 KB, a little more than the operand images: a workgroup of the look-ahead panel chain (leaf / chain kernel, 73 KB) must
    // fit into the LDS range one retiring workgroup of this kernel frees
    __shared__ double smem[2][2][IMG + 128];
    const int tid = threadIdx.x;
    int ti, tj;
    if (g.tab) {                      // balanced table: every XCD (blocks b, b + 8, ..) gets the same number of real tiles
        const int e = g.tab[blockIdx.x];
        if (e < 0) return;
        ti = e >> 16; tj = e & 0xffff;
    } else {
        const long t = xcd_remap(blockIdx.x, gridDim.x, g.tiles_n);
        if (g.lower == 2) tile_of_rs(t, g.tiles_m, g.tiles_n, g.ls, g.lo, ti, tj);
        else tile_of(t, g.tiles_m, g.tiles_n, g.lower == 1, ti, tj);
        if (ti >= g.tiles_m || tj >= g.tiles_n) return;
        if (g.lower == 1 && tj > ti) return;
        if (g.lower == 2 && tj > ti * g.ls + g.lo) return;
    }
    if (g.rev) ti = g.tiles_m - 1 - ti;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    long kbeg = g.kb0 + g.kbi * ti + g.kbj * tj;
    long kend = (g.ke0 < 0 ? g.K : g.ke0 + g.kei * ti + g.kej * tj);
    if (kbeg < 0) kbeg = 0;
    if (kend > g.K) kend = g.K;
    long coff = 0;
    const double *gA = g.A, *gB = g.B;
    if (g.ksplit) {
        if (g.tri) {
            const long kt = 128L * (tj + 1);
            if ((long)blockIdx.y * g.ksplit >= kt) return;
            if (kend > kt) kend = kt;
        }
        kbeg += (long)blockIdx.y * g.ksplit;
        if (kbeg + g.ksplit < kend) kend = kbeg + g.ksplit;
        coff = (long)blockIdx.y * g.csplit;
    } else if (g.ny) {
        const long y = blockIdx.y % g.ny, z = blockIdx.y / g.ny;
        gA += y * g.ab1 + z * g.ab2; gB += y * g.bb1 + z * g.bb2; coff = y * g.cb1 + z * g.cb2;
    }
    const int nk = kend > kbeg ? (int)((kend - kbeg) / BK) : 0;
    const long m0 = (long)ti * 128, n0 = (long)tj * 128;
    long nb0 = n0;                    // first row of this tile's B block
    if (!BNM) { const int idx = tj + g.bco; nb0 = ((long)(idx % g.bcr) * g.bcb + idx / g.bcr) * 128; }
    // both operands k-minor: unpadded [128][16] images whose 16-byte chunks are XOR-swizzled within their row by
    // s(row) = bit1(row) | bit2(row) << 2.  ds_read_b128 serves a wave in four groups of sixteen lanes, {0-3,12-15,20-27},
    // {4-11,16-19,28-31} and the same +32: every group holds each r = lane & 15 once, with lane group q = lane >> 4 alternating
    // between two values, and with this swizzle the sixteen chunks of a group fall on sixteen different 16-byte bank groups;
    // a row is still 128 contiguous bytes for the fill.
    constexpr bool KM = !AKM && !BNM;
    auto swz = [](int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); };
    // global -> register staging maps of the first K step (4 x 16 B per operand per thread)
    const double *ga[4]; const double *gb[4];
    int sa[4], sb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (AKM) {           // A stored (K, M): rows of 128 contiguous m
            int kr = p * 4 + (tid >> 6), mc = (tid & 63) * 2;
            ga[p] = gA + (kbeg + kr) * g.lda + m0 + mc;
            sa[p] = kr * LDM + mc;
        } else {             // A stored (M, K): rows of 16 contiguous k
            int row = p * 32 + (tid >> 3), kc = (tid & 7) * 2;
            ga[p] = gA + (m0 + row) * g.lda + kbeg + kc;
            sa[p] = KM ? row * 16 + (((tid & 7) ^ swz(row)) << 1) : row * LDK + kc;
        }
        if (BNM) {           // B stored (K, N)
            int kr = p * 4 + (tid >> 6), nc = (tid & 63) * 2;
            gb[p] = gB + (kbeg + kr) * g.ldb + n0 + nc;
            sb[p] = kr * LDM + nc;
        } else {             // B stored (N, K)
            int row = p * 32 + (tid >> 3), kc = (tid & 7) * 2;
            gb[p] = gB + (nb0 + row) * g.ldb + kbeg + kc;
            sb[p] = KM ? row * 16 + (((tid & 7) ^ swz(row)) << 1) : row * LDK + kc;
        }
    }
    const long astep = AKM ? (long)BK * g.lda : BK;
    const long bstep = BNM ? (long)BK * g.ldb : BK;
    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double2_t ra[4], rb[4];
    if (nk > 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = *reinterpret_cast<const double2_t *>(ga[p]);
            rb[p] = *reinterpret_cast<const double2_t *>(gb[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<double2_t *>(&smem[0][0][sa[p]]) = ra[p];
            *reinterpret_cast<double2_t *>(&smem[0][1][sb[p]]) = rb[p];
        }
    }
    __syncthreads();
    if constexpr (KM) {
        // K loop without vector-ALU work.  fp64 MFMA and the vector ALU do not execute side by side on a SIMD
        // (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in this kernel): every pointer bump, every LDS base recomputed per step comes straight
        // out of the MFMA stream.  The global loads go through buffer descriptors -- a constant 32-bit offset per lane, the K
        // position in an SGPR bumped by the scalar ALU -- and write the swizzled image directly (LDS-DMA: a wave instruction
        // fills eight whole rows, lane l lands 16 l bytes behind the wave's base, so it fetches chunk (l & 7) ^ s(row) of row
        // l >> 3 of its rows); the loop is unrolled over the two LDS buffers so that every LDS address is one loop-invariant
        // register plus an immediate.
        int fa[4], fb[4], fa1[4], fb1[4];               // second half of a K step = the neighbouring chunk (chunk ^ 1)
        {
            const int c0 = (2 * q) ^ swz(r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = (wm * 64 + i * 16 + r) * 16 + 2 * c0; fa1[i] = (wm * 64 + i * 16 + r) * 16 + 2 * (c0 ^ 1);
                fb[i] = (wn * 64 + i * 16 + r) * 16 + 2 * c0; fb1[i] = (wn * 64 + i * 16 + r) * 16 + 2 * (c0 ^ 1);
            }
        }
        const double *abase = uniform_ptr(gA + m0 * g.lda + kbeg);
        const double *bbase = uniform_ptr(gB + nb0 * g.ldb + kbeg);
        const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(abase), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(bbase), 0, 0xffffffff, 0x00020000);
        int voa[4], vob[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = p * 32 + (tid >> 3), kc = ((tid & 7) ^ swz(row)) * 2;
            voa[p] = (int)(((long)row * g.lda + kc) * 8);
            vob[p] = (int)(((long)row * g.ldb + kc) * 8);
        }
        typedef __attribute__((address_space(3))) void lds_void;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // the LDS destination of a wave's DMA is a scalar (M0)
        auto dma_step = [&](auto bufc, int so) {
            constexpr int BUF = decltype(bufc)::value;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                lds_void *da = (lds_void *)&smem[BUF][0][(p * 32 + wave_u * 8) * 16];
                lds_void *db = (lds_void *)&smem[BUF][1][(p * 32 + wave_u * 8) * 16];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, da, 16, voa[p], so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, db, 16, vob[p], so, 0, 0);
            }
        };
        int soff = 0;                                   // byte offset of the K step being fetched
        // Cooperative yield (the trailing update only).  A workgroup of the panel chain that shares this compute unit
        // (look-ahead) is a chain of dependent vector / MFMA instructions, and beside this loop each of them waits for a
        // 64-cycle MFMA: 130-250 us for a leaf instead of 37.  It raises its CU's counter; every wave here reads that word once
        // per K step with a scalar load (no vector-ALU work: issued behind the step's LDS reads, long landed when the step's
        // MFMAs are through) and sleeps while it is up.  Bounded: a wave sleeps at most ~1 ms per tile whatever the counter says.
        constexpr bool YIELD = ROLE == 1;
        const int *yp = nullptr;
        int ybudget = 256;
        if constexpr (YIELD) yp = cu_yield_slot(const_cast<int *>(g.yield));
        auto kstep = [&](auto curc, const bool more) {
            constexpr int CUR = decltype(curc)::value;
            if (more) {
                soff += BK * 8;
                dma_step(std::integral_constant<int, CUR ^ 1>{}, soff);
            }
            const double *pa = &smem[CUR][0][0];
            const double *pb = &smem[CUR][1][0];
            __builtin_amdgcn_s_setprio(ROLE ? 1 : 2);
            // all sixteen fragment reads of the step go out before its first MFMA
            double2_t a2[2][4], b2[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a2[hf][i] = *reinterpret_cast<const double2_t *>(pa + (hf ? fa1[i] : fa[i]));
                    b2[hf][i] = *reinterpret_cast<const double2_t *>(pb + (hf ? fb1[i] : fb[i]));
                }
            __builtin_amdgcn_sched_barrier(0);
            int yv = 0;
            if constexpr (YIELD) asm volatile("s_load_dword %0, %1, 0x0 glc" : "={s95}"(yv) : "s"(yp));
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[hf][i][s], b2[hf][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(ROLE ? 0 : 1);
            if constexpr (YIELD) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+{s95}"(yv) :: "memory");
                while (yv != 0 && ybudget > 0) {
                    --ybudget;
                    __builtin_amdgcn_s_sleep(127);
                    asm volatile("s_load_dword %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "={s95}"(yv) : "s"(yp) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next step's image has landed
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            kstep(std::integral_constant<int, 0>{}, true);
            kstep(std::integral_constant<int, 1>{}, kt + 2 < nk);
        }
        if (kt < nk) kstep(std::integral_constant<int, 0>{}, false);
    } else {
        // (K,M) and / or (K,N) operands: global -> registers -> LDS, padded images, 8-byte fragment reads
        int fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = AKM ? (q * LDM + wm * 64 + i * 16 + r) : ((wm * 64 + i * 16 + r) * LDK + q);
            fb[i] = BNM ? (q * LDM + wn * 64 + i * 16 + r) : ((wn * 64 + i * 16 + r) * LDK + q);
        }
        constexpr int SA = AKM ? 4 * LDM : 4;
        constexpr int SB = BNM ? 4 * LDM : 4;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < nk);
            if (more) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    ga[p] += astep; gb[p] += bstep;
                    ra[p] = *reinterpret_cast<const double2_t *>(ga[p]);
                    rb[p] = *reinterpret_cast<const double2_t *>(gb[p]);
                }
            }
            const double *pa = &smem[cur][0][0];
            const double *pb = &smem[cur][1][0];
            // the fragment reads and MFMAs of this wave go out at raised priority; the memory phase of the K step (global
            // loads above, LDS writes and barrier below) yields to the co-resident workgroup's MFMAs
            __builtin_amdgcn_s_setprio(ROLE ? 1 : 2);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double a[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = pa[fa[i] + s * SA]; bv[i] = pb[fb[i] + s * SB]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bv[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(ROLE ? 0 : 1);
            if (more) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    *reinterpret_cast<double2_t *>(&smem[cur ^ 1][0][sa[p]]) = ra[p];
                    *reinterpret_cast<double2_t *>(&smem[cur ^ 1][1][sb[p]]) = rb[p];
                }
            }
            __syncthreads();
        }
    }
    store_tile(acc, g.C + coff + (m0 + wm * 64 + q) * g.ldc + n0 + wn * 64 + r, g.ldc, g.alpha, g.beta);
}

// Small-tile variant for the latency-bound steps of the panel chain (TRSM by the inverted diagonal block, in-panel
// update with K = 128) when a launch has too few 128 x 128 tiles to fill the chip: TM x TN tiles of a quarter the size,
// so four times the workgroups share the same product and a K = 128 tile is 16 MFMAs per wave instead of 64.
// (M,K) x (N,K) layout only, plain K range, plain tile grid (a launch this small has nothing to gain from the
// XCD-aware order).  <64, 64> for products whose result does not alias an operand; <32, 128> for the in-place TRSM
// (C is A): a workgroup then owns whole rows, so nobody overwrites operand columns another workgroup still reads;
// <128, 32> likewise for C = inv(L_kk) B in place of B (K, N): a workgroup owns whole columns.  BNM: B stored (K, N).
template <int TM, int TN, int BNM>
__global__ __launch_bounds__(256, 2) void gemm_f64_small_kernel(GemmArgs g) {
    constexpr int WN = TN / 32, WM = TM / 32;          // 4 waves of 32 x 32 each
    static_assert(WM * WN == 4, "four waves of 32 x 32");
    constexpr int PA = TM / 32;                        // load passes of 256 threads x 16 bytes: A image (TM x 16, k-minor)
    constexpr int PB = TN / 32;                        //   B image: (TN x 16, k-minor) or, BNM, (16 x TN, n-minor)
    constexpr int LDN = TN + 16;                       // row stride of the n-minor B image (16 mod 32 doubles: conflict-free reads)
    constexpr int IMA = TM * LDK, IMB = BNM ? 16 * LDN : TN * LDK;
    __shared__ double smem[2][IMA + IMB];
    __builtin_amdgcn_s_setprio(2);                     // chain steps and sub-round updates: ahead of the trailing update's waves
    const int tn = (g.tiles_n * 128) / TN;             // tiles per row of the tile grid
    const int ti = blockIdx.x / tn, tj = blockIdx.x % tn;
    const long m0 = (long)ti * TM, n0 = (long)tj * TN;
    if (g.lower == 1 && n0 / 128 > m0 / 128) return;   // whole 128-tiles on / below the diagonal, as the ABI says
    if (g.lower == 2 && n0 / 128 > (m0 / 128) * g.ls + g.lo) return;        // row-sharded trailing update: this rank's block rows
    const int tid = threadIdx.x;
    int *yflag = nullptr;             // a latency-bound step of the chain: the co-resident trailing-update workgroup sleeps meanwhile
    if (g.raise && tid == 0) { yflag = cu_yield_slot(g.raise); atomicAdd(yflag, 1); }
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;
    const int nk = (int)(g.K / BK);

    const double *ga[PA]; const double *gb[PB];
    int sa[PA], sb[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = p * 32 + (tid >> 3), kc = (tid & 7) * 2;
        ga[p] = g.A + (m0 + row) * g.lda + kc;
        sa[p] = row * LDK + kc;
    }
    constexpr int TPR = TN / 2;                         // BNM: threads per k-row of the B image
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        if (BNM) {
            const int kr = p * (256 / TPR) + tid / TPR, nc = (tid % TPR) * 2;
            gb[p] = g.B + (long)kr * g.ldb + n0 + nc;
            sb[p] = IMA + kr * LDN + nc;
        } else {
            // B rows in all-gather (block-cyclic) order, as in the 128-tile kernel: 128-row block `idx` of the cyclic order
            const int row = p * 32 + (tid >> 3), kc = (tid & 7) * 2;
            const int idx = (int)(n0 / 128) + g.bco;
            const long nb0 = ((long)(idx % g.bcr) * g.bcb + idx / g.bcr) * 128 + n0 % 128;
            gb[p] = g.B + (nb0 + row) * g.ldb + kc;
            sb[p] = IMA + row * LDK + kc;
        }
    }
    const long bstep = BNM ? (long)BK * g.ldb : BK;
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = (wm * 32 + i * 16 + r) * LDK + q;
        fb[i] = IMA + (BNM ? (q * LDN + wn * 32 + i * 16 + r) : ((wn * 32 + i * 16 + r) * LDK + q));
    }
    constexpr int SB = BNM ? 4 * LDN : 4;
    double4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // the read-modify-write operand of the epilogue is fetched up front: its latency hides under the K loop
    double *cbase = g.C + (m0 + wm * 32 + q) * g.ldc + n0 + wn * 32 + r;
    double old[2][4][2];
    if (g.beta != 0.0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int j = 0; j < 2; ++j) old[i][v][j] = cbase[(i * 16 + 4 * v) * g.ldc + j * 16];
    }
    double2_t ra[PA], rb[PB];
    if (nk > 0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const double2_t *>(ga[p]);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const double2_t *>(gb[p]);
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<double2_t *>(&smem[0][sa[p]]) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<double2_t *>(&smem[0][sb[p]]) = rb[p];
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int p = 0; p < PA; ++p) { ga[p] += BK; ra[p] = *reinterpret_cast<const double2_t *>(ga[p]); }
#pragma unroll
            for (int p = 0; p < PB; ++p) { gb[p] += bstep; rb[p] = *reinterpret_cast<const double2_t *>(gb[p]); }
        }
        const double *ps = &smem[cur][0];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            double a[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[i] = ps[fa[i] + s4 * 4]; bv[i] = ps[fb[i] + s4 * SB]; }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int p = 0; p < PA; ++p) *reinterpret_cast<double2_t *>(&smem[cur ^ 1][sa[p]]) = ra[p];
#pragma unroll
            for (int p = 0; p < PB; ++p) *reinterpret_cast<double2_t *>(&smem[cur ^ 1][sb[p]]) = rb[p];
        }
        __syncthreads();
    }
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                cbase[(i * 16 + 4 * v) * g.ldc + j * 16] = alpha * acc[i][j][v] + (beta != 0.0 ? beta * old[i][v][j] : 0.0);
    if (yflag) atomicAdd(yflag, -1);
}

// K = 128 in ONE stage, for the two products every 128-column step of the panel chain waits for (TRSM by the inverted
// diagonal block, update of the next block column): the whole k range of both operands is fetched at once -- every load
// of the workgroup in flight together -- instead of eight double-buffered steps of one memory round trip each (the
// steps are latency, not bandwidth: these launches have a few hundred workgroups).  (M,K) x (N,K) layout, k-minor images
// with a row stride of 130 doubles (16-byte fragment reads of the permuted k order, conflict-free).  A workgroup owns 32
// rows: its A image stays, the B rows pass through in TN / 32 phases of 32 (the next phase's loads fly during the MFMAs),
// a wave computes one 16 x 16 tile per phase.  67 KB of LDS -- like the leaf, it must fit into what one retiring
// trailing-update workgroup frees.  <64>: results that alias no operand; <128>: the in-place TRSM (the workgroup has read
// all of its rows before it writes them).
template <int TN>
__device__ __forceinline__ void k128_chunk(const GemmArgs &g, const int vb, double *sA, double *sB) {
    constexpr int LDS_ = 130, NPH = TN / 32, TM = 32;
    const int tn = (g.tiles_n * 128) / TN;
    const int ti = vb / tn, tj = vb % tn;
    const long m0 = (long)ti * TM, n0 = (long)tj * TN;
    if (g.lower == 1 && n0 / 128 > m0 / 128) return;
    if (g.lower == 2 && n0 / 128 > (m0 / 128) * g.ls + g.lo) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int idx = (int)(n0 / 128) + g.bco;                            // B rows in all-gather (block-cyclic) order
    const long nb0 = ((long)(idx % g.bcr) * g.bcb + idx / g.bcr) * 128 + n0 % 128;

    // one wave reads one row of 128 doubles per pass
    constexpr int PP = 8;
    double2_t ra[PP], rb[PP];
    const int lrow = wave, kc = lane * 2;
#pragma unroll
    for (int p = 0; p < PP; ++p) ra[p] = *reinterpret_cast<const double2_t *>(g.A + (m0 + p * 4 + lrow) * g.lda + kc);
#pragma unroll
    for (int p = 0; p < PP; ++p) rb[p] = *reinterpret_cast<const double2_t *>(g.B + (nb0 + p * 4 + lrow) * g.ldb + kc);
    double *cbase = g.C + (m0 + wm * 16 + q) * g.ldc + n0 + wn * 16 + r;
    double old[NPH][4];
    if (g.beta != 0.0) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int v = 0; v < 4; ++v) old[ph][v] = cbase[(4 * v) * g.ldc + ph * 32];
    }
#pragma unroll
    for (int p = 0; p < PP; ++p) *reinterpret_cast<double2_t *>(&sA[(p * 4 + lrow) * LDS_ + kc]) = ra[p];
    double4_t acc[NPH];
    const double *pa = &sA[(wm * 16 + r) * LDS_ + 4 * q];
    const double *pb = &sB[(wn * 16 + r) * LDS_ + 4 * q];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
#pragma unroll
        for (int p = 0; p < PP; ++p) *reinterpret_cast<double2_t *>(&sB[(p * 4 + lrow) * LDS_ + kc]) = rb[p];
        __syncthreads();
        if (ph + 1 < NPH) {                                             // the next 32 rows of B are on their way during the MFMAs
#pragma unroll
            for (int p = 0; p < PP; ++p)
                rb[p] = *reinterpret_cast<const double2_t *>(g.B + (nb0 + (ph + 1) * 32 + p * 4 + lrow) * g.ldb + kc);
        }
        // two accumulators per tile (even / odd halves of the k range): the MFMAs of a tile do not wait for each other
        double4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const double2_t a0 = *reinterpret_cast<const double2_t *>(pa + kb * 16);
            const double2_t a1 = *reinterpret_cast<const double2_t *>(pa + kb * 16 + 2);
            const double2_t b0 = *reinterpret_cast<const double2_t *>(pb + kb * 16);
            const double2_t b1 = *reinterpret_cast<const double2_t *>(pb + kb * 16 + 2);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], c1, 0, 0, 0);
        }
        acc[ph] = c0 + c1;
        if (ph + 1 < NPH) __syncthreads();                              // everybody is done with this B image
    }
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            cbase[(4 * v) * g.ldc + ph * 32] = alpha * acc[ph][v] + (beta != 0.0 ? beta * old[ph][v] : 0.0);
}

template <int TN>
__global__ __launch_bounds__(256) void gemm_f64_k128_kernel(GemmArgs g) {
    constexpr int LDS_ = 130, TM = 32;
    __shared__ double sA[TM * LDS_];
    __shared__ double sB[32 * LDS_];
    __builtin_amdgcn_s_setprio(3);        // a step of the chain: ahead of the trailing-update waves it shares its SIMDs with
    int *yflag = nullptr;                 // the co-resident trailing-update workgroup sleeps while this one runs (10 us instead of 53)
    if (g.raise && threadIdx.x == 0) { yflag = cu_yield_slot(g.raise); atomicAdd(yflag, 1); }
    k128_chunk<TN>(g, (int)blockIdx.x, sA, sB);
    if (yflag) atomicAdd(yflag, -1);
}

// The chain's panel TRSM without the inverse of the 128 x 128 diagonal block: X = A inv(L)^T by substitution over the eight
// 16-column tiles,  X_t = (A_t - sum_{s<t} X_s L[t,s]^T) inv(L_tt)^T,  with the inverses of the 16 x 16 diagonal tiles only
// (what the leaf computes anyway; its triangular inverse of the whole block was 19 of its 103 thousand cycles, on ONE
// workgroup, in every step of the chain -- here the same number of MFMAs is spread over rows/32 workgroups).  A workgroup
// owns 32 rows, a wave 16 of them; everything is computed transposed so that a finished tile is already the next
// product's B operand: R^T = A_t^T - sum L[t,s] X_s^T accumulates in the MFMA D layout, X_t^T = inv(L_tt) R^T takes it as
// it is.  L passes through LDS in four phases of 32 rows (all of it is fetched into registers up front); 67 KB of LDS.
struct TrsmTilesArgs { double *A; long lda; const double *L; long ldl; const double *dinv; int *raise; };

__device__ __forceinline__ void trsm_tiles_chunk(const TrsmTilesArgs &g, const int vb, double *sX, double *sL) {
    constexpr int LDS_ = 130;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    double *Arows = g.A + (long)vb * 32 * g.lda;
    // every global load of the kernel goes out at once (the phases would otherwise each wait a memory round trip):
    // the workgroup's 32 rows of A, all 128 rows of L (four phases of 32), the eight tile inverses
    // (two of the four phases of L up front, the other two as soon as a phase's registers have gone to LDS -- a phase computes
    // for longer than a load takes: 245 registers instead of 362, see the launch bounds)
    // Addresses: one buffer descriptor per operand, the lane's 16 bytes as the only vector offset, the row as a scalar offset
    // (a 64-bit pointer per load would hold 112 more registers than the data).
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Arows)), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t l_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(g.L)), 0, 0xffffffff, 0x00020000);
    const int vo = 16 * lane;
    const int arow = (int)(g.lda * 8), lrow = (int)(g.ldl * 8);
    u32x4 ra[16], rl[2][16];
    double2_t rd[8];
#pragma unroll
    for (int p = 0; p < 16; ++p) ra[p] = __builtin_amdgcn_raw_buffer_load_b128(a_src, vo, (2 * p + wave_u) * arow, 0);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int p = 0; p < 16; ++p) rl[ph][p] = __builtin_amdgcn_raw_buffer_load_b128(l_src, vo, (32 * ph + 2 * p + wave_u) * lrow, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) rd[t] = *reinterpret_cast<const double2_t *>(g.dinv + t * 256 + 2 * tid);
#pragma unroll
    for (int p = 0; p < 16; ++p) *reinterpret_cast<u32x4 *>(&sX[(2 * p + wave) * LDS_ + 2 * lane]) = ra[p];
    double *xrow = &sX[(16 * wave + r) * LDS_];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
        for (int p = 0; p < 16; ++p) *reinterpret_cast<u32x4 *>(&sL[(2 * p + wave) * LDS_ + 2 * lane]) = rl[ph & 1][p];
        if (ph + 2 < 4) {
            __builtin_amdgcn_sched_barrier(0);        // behind the LDS writes: the phase's registers are free again
#pragma unroll
            for (int p = 0; p < 16; ++p) rl[ph & 1][p] = __builtin_amdgcn_raw_buffer_load_b128(l_src, vo, (32 * (ph + 2) + 2 * p + wave_u) * lrow, 0);
        }
        __syncthreads();
        {   // the two diagonal tiles of these rows <- their inverses (thread -> two adjacent entries of each tile)
            const int a = tid >> 3, b = (tid & 7) * 2;
            *reinterpret_cast<double2_t *>(&sL[a * LDS_ + 32 * ph + b]) = rd[2 * ph];
            *reinterpret_cast<double2_t *>(&sL[(16 + a) * LDS_ + 32 * ph + 16 + b]) = rd[2 * ph + 1];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = 2 * ph + h;
            const double *lrow = &sL[(16 * h + r) * LDS_];
            double4_t acc, acc2 = {0.0, 0.0, 0.0, 0.0}, acc3 = {0.0, 0.0, 0.0, 0.0}, acc4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = xrow[16 * t + q + 4 * v];
            // sum over the finished tiles: lane group q takes k = 4q .. 4q+3 of a tile (same permutation on both operands: two
            // 16-byte reads per operand and tile), four accumulators so that no MFMA waits for the one before it
#pragma unroll
            for (int s = 0; s < t; ++s) {
                const double2_t l01 = *reinterpret_cast<const double2_t *>(lrow + 16 * s + 4 * q);
                const double2_t l23 = *reinterpret_cast<const double2_t *>(lrow + 16 * s + 4 * q + 2);
                const double2_t x01 = *reinterpret_cast<const double2_t *>(xrow + 16 * s + 4 * q);
                const double2_t x23 = *reinterpret_cast<const double2_t *>(xrow + 16 * s + 4 * q + 2);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[0], x01[0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[1], x01[1], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[0], x23[0], acc3, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[1], x23[1], acc4, 0, 0, 0);
            }
            acc += (acc2 + acc3) + acc4;
            double4_t x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lrow[16 * t + q], acc[0], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lrow[16 * t + 4 + q], acc[1], x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lrow[16 * t + 8 + q], acc[2], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lrow[16 * t + 12 + q], acc[3], x1, 0, 0, 0);
            x0 += x1;
#pragma unroll
            for (int v = 0; v < 4; ++v) xrow[16 * t + q + 4 * v] = x0[v];
        }
        if (ph + 1 < 4) __syncthreads();          // both waves are done with these rows of L
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p)      // (row offset in the VECTOR offset: a 16-byte store with an SGPR offset gets no wait state before its data registers are reused, chain.hip trsm_sub)
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&sX[(2 * p + wave) * LDS_ + 2 * lane]), a_src, vo + (2 * p + wave) * arow, 0, 0);
}

// 236 VGPR + 54 AGPR = 296 registers per lane (two of the four phases of L in registers at a time: 362 with all four), i.e. one
// wave per SIMD: beside a trailing update whose waves hold 224 each a workgroup starts once BOTH update workgroups of a compute
// unit have retired.  Used by the launch-per-step chain only -- panels with fewer than 4096 rows below them, where no large
// update runs beside the chain, and the row-sharded driver's stacked panel; the resident panel kernel (chain.hip) solves by
// trsm_sub at <= 128 registers.
__global__ __launch_bounds__(128) void trsm_tiles_kernel(TrsmTilesArgs g) {
    constexpr int LDS_ = 130;
    __shared__ double sX[32 * LDS_];
    __shared__ double sL[32 * LDS_];
    __builtin_amdgcn_s_setprio(3);        // a step of the chain: ahead of the trailing-update waves it shares its SIMDs with
    int *yflag = nullptr;                 // the co-resident trailing-update workgroup sleeps while this one runs (10 us instead of 53)
    if (g.raise && threadIdx.x == 0) { yflag = cu_yield_slot(g.raise); atomicAdd(yflag, 1); }
    trsm_tiles_chunk(g, (int)blockIdx.x, sX, sL);
    if (yflag) atomicAdd(yflag, -1);
}

__global__ void mfma_selftest_kernel(const double *A, const double *B, double *D) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    // A is 16x4 row-major, B is 4x16 row-major
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r * 4 + q], B[q * 16 + r], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) D[(q + 4 * v) * 16 + r] = acc[v];
}

// register-only MFMA stream (no memory traffic): the ceiling the chip sustains for this
// instruction mix under its own clock management; used by tools/ and DESIGN.md, not by the path
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(double *out, int iters, double seed) {
    double4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = seed * (threadIdx.x + i + 1) * 1e-3; b[i] = seed * (threadIdx.x * 3 + i + 7) * 1e-3; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

int launch_mfma_peak(fvgp_handle *h, double *out, int blocks, int iters) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, out, iters, 1.0);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_mfma_selftest(fvgp_handle *h, const double *A, const double *B, double *D) {
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, h->stream, A, B, D);
    HIPCHK(hipGetLastError());
    return 0;
}

static long gemm_grid_tiles(int tiles_m, int tiles_n, int lower) {
    // whole super-tiles (out-of-range tiles exit at once); must mirror tile_of()
    const long S = 8;
    const long SN = tiles_n < S ? tiles_n : S;
    const long sm = (tiles_m + S - 1) / S, sn = (tiles_n + SN - 1) / SN;
    long nst;
    if (lower && SN == S) {
        if (sm <= sn) nst = sm * (sm + 1) / 2;
        else nst = sn * (sn + 1) / 2 + (sm - sn) * sn;
    } else nst = sm * sn;
    return nst * S * SN;
}

static long gemm_grid_tiles_rs(int tiles_m, int tiles_n, int ls, int lo) {
    const int S = 8, SN = tiles_n < S ? tiles_n : S;
    long nst = 0;
    for (int si = 0; si < (tiles_m + S - 1) / S; ++si) nst += rs_super_count(si, tiles_n, SN, ls, lo);
    return nst * S * SN;
}

// host-side replay of the blockIdx -> tile map (XCD remap included) for the CPU tests
long gemm_debug_tile_map(int tiles_m, int tiles_n, int lower, int ls, int lo, int *out_ti, int *out_tj, long cap) {
    const long nwg = lower == 2 ? gemm_grid_tiles_rs(tiles_m, tiles_n, ls, lo) : gemm_grid_tiles(tiles_m, tiles_n, lower);
    for (long b = 0; b < nwg && b < cap; ++b) {
        const long t = xcd_remap(b, nwg, tiles_n);
        int ti, tj;
        if (lower == 2) tile_of_rs(t, tiles_m, tiles_n, ls, lo, ti, tj);
        else tile_of(t, tiles_m, tiles_n, lower, ti, tj);
        out_ti[b] = ti; out_tj[b] = tj;
    }
    return nwg;
}

// ---------------------------------------------------------------------------------------
// Balanced tile tables.  The formula map deals whole 8 x 8 super-tiles to the XCDs; diagonal super-tiles are half empty and
// the last ones partial, so the XCD with the most real tiles sets a launch's time: +2.5 % at 130-200 tile rows, +8-10 % at
// <= 100, +6-25 % on the row-sharded (lower == 2) grids of an 8-rank run.  The table keeps the same super-tile walk (the 64
// tiles of a super-tile run together on one L2) but cuts the sequence of REAL tiles into eight equal contiguous runs, one per
// XCD (block b -> XCD b % 8, entry b / 8 of its run).  Built on the host once per launch shape, kept on the device.
static void drop_tile_tables(fvgp_handle *h) {
    for (auto &kv : h->tile_tabs) { (void)hipFree(kv.second.dev); if (kv.second.host) (void)hipHostFree(kv.second.host); }
    h->tile_tabs.clear();
    h->tile_tab_bytes = 0;
}

void gemm_release_tables(fvgp_handle *h) {
    drop_tile_tables(h);
    if (h->copy_stream) { (void)hipStreamDestroy(h->copy_stream); h->copy_stream = nullptr; }
}

static std::vector<int> build_tile_table(int tm, int tn, int lower, int ls, int lo) {
    // real tiles in the formula map's own super-tile order
    const long nform = lower == 2 ? gemm_grid_tiles_rs(tm, tn, ls, lo) : gemm_grid_tiles(tm, tn, lower == 1);
    std::vector<int> real;
    real.reserve((size_t)nform);
    for (long t = 0; t < nform; ++t) {
        int ti, tj;
        if (lower == 2) tile_of_rs(t, tm, tn, ls, lo, ti, tj);
        else tile_of(t, tm, tn, lower == 1, ti, tj);
        if (ti >= tm || tj >= tn) continue;
        if (lower == 1 && tj > ti) continue;
        if (lower == 2 && tj > (long)ti * ls + lo) continue;
        real.push_back((ti << 16) | tj);
    }
    const long T = (long)real.size();
    const long per = (T + 7) / 8;
    std::vector<int> tab((size_t)(per * 8), -1);
    for (long x = 0; x < 8; ++x) {                 // XCD x takes the contiguous run [x*T/8, (x+1)*T/8)
        const long a = x * T / 8, b = (x + 1) * T / 8;
        for (long i = a; i < b; ++i) tab[(size_t)((i - a) * 8 + x)] = real[(size_t)i];
    }
    return tab;
}

// host-side view of the table for the CPU tests: entries (ti << 16) | tj or -1, returns the grid size
long gemm_debug_tile_table(int tiles_m, int tiles_n, int lower, int ls, int lo, int *out, long cap) {
    const std::vector<int> tab = build_tile_table(tiles_m, tiles_n, lower, ls, lo);
    for (size_t i = 0; i < tab.size() && (long)i < cap; ++i) out[i] = tab[i];
    return (long)tab.size();
}

static int tile_table(fvgp_handle *h, int tm, int tn, int lower, int ls, int lo, const int **dev, long *grid) {
    const TileTabKey key{tm, tn, lower, ls, lo};
    auto it = h->tile_tabs.find(key);
    if (it == h->tile_tabs.end()) {
        const std::vector<int> tab = build_tile_table(tm, tn, lower, ls, lo);
        TileTab tt{nullptr, (long)tab.size(), nullptr};
        const size_t bytes = (size_t)tt.grid * sizeof(int);
        // a process that keeps growing its problem (active learning: np + 128 per step) meets new launch shapes for ever: the
        // cache is dropped as a whole beyond 256 MB (every queued launch has finished with its table after the device sync)
        if (h->tile_tab_bytes + bytes > ((size_t)256 << 20)) {
            HIPCHK(hipDeviceSynchronize());
            drop_tile_tables(h);
        }
        if (tt.grid > 0) {
            // uploaded from pinned memory on a stream of its own, and the host waits for THAT stream: no null-stream copy that
            // would join the compute streams (look-ahead, the enqueue-only entry points)
            if (!h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
            HIPCHK(hipMalloc((void **)&tt.dev, bytes));
            HIPCHK(hipHostMalloc((void **)&tt.host, bytes, hipHostMallocDefault));
            memcpy(tt.host, tab.data(), bytes);
            HIPCHK(hipMemcpyAsync(tt.dev, tt.host, bytes, hipMemcpyHostToDevice, h->copy_stream));
            HIPCHK(hipStreamSynchronize(h->copy_stream));
            h->tile_tab_bytes += bytes;
        }
        it = h->tile_tabs.emplace(key, tt).first;
    }
    *dev = it->second.dev; *grid = it->second.grid;
    return 0;
}

// a trailing update that small -- under about one round of 128-tiles -- is a step of the chain too: the columns of the
// next panel wait for it
bool gemm_takes_small_tiles(const fvgp_handle *h, const GemmDesc &d) {
    const long t128 = (long)(d.M / 128) * (d.N / 128);
    const bool few = d.role == 1 ? (t128 <= h->small_tile_max_update) : (t128 <= h->small_tile_max && d.K <= 512);
    if (d.b_nmajor && ((const double *)d.C == d.A || ((const double *)d.C == d.B && d.M != 128))) return false;
    // (N,K) layout in place: a small-tile workgroup owns 32 whole rows of C.  C == B would overwrite rows of B that other
    // workgroups still read as their operand, and C == A is only safe when those rows are ALL of the tile's columns (N == 128)
    if (!d.b_nmajor && (const double *)d.C == d.B) return false;
    if (!d.b_nmajor && (const double *)d.C == d.A && d.N != 128) return false;
    if (d.b_nmajor && (d.lower == 2 || d.bc_ranks != 1 || d.bc_off != 0)) return false;
    return !d.a_kmajor && !d.rev_m && few && d.kb0 == 0 && d.kbi == 0 && d.kbj == 0 && d.ke0 < 0;
}

int launch_gemm(fvgp_handle *h, const GemmDesc &d) {
    if (d.M <= 0 || d.N <= 0) return 0;
    if (d.M % 128 || d.N % 128 || d.K % BK || d.K < 0) { fvgp_set_error("gemm: M,N must be multiples of 128 and K of 16"); return -5; }
    if ((d.lda & 1) || (d.ldb & 1) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) {
        fvgp_set_error("gemm: operands must be 16-byte aligned with even leading dimensions"); return -9;
    }
    if (d.lda >= (1L << 21) || d.ldb >= (1L << 21)) {     // a tile's 128 rows are addressed by 32-bit byte offsets from its first row
        fvgp_set_error("gemm: leading dimensions must be below 2^21 doubles"); return -9;
    }
    GemmArgs g;
    g.A = d.A; g.B = d.B; g.C = d.C; g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.alpha = d.alpha; g.beta = d.beta; g.K = d.K;
    g.tiles_m = (int)(d.M / 128); g.tiles_n = (int)(d.N / 128); g.lower = d.lower; g.ls = d.lower_scale; g.lo = d.lower_off;
    g.bcr = d.bc_ranks; g.bcb = d.bc_blocks; g.bco = d.bc_off; g.rev = d.rev_m;

    if (d.rev_m && d.lower) { fvgp_set_error("gemm: rev_m is for full (non-triangular) tile grids"); return -3; }
    if (d.bc_ranks < 1) return -7;
    if ((d.bc_ranks > 1 || d.bc_off) && d.b_nmajor) { fvgp_set_error("gemm: block-cyclic B needs the (N, K) layout"); return -7; }
    g.kb0 = d.kb0; g.kbi = d.kbi; g.kbj = d.kbj; g.ke0 = d.ke0; g.kei = d.kei; g.kej = d.kej;
    g.ntiles = g.lower == 2 ? gemm_grid_tiles_rs(g.tiles_m, g.tiles_n, g.ls, g.lo) : gemm_grid_tiles(g.tiles_m, g.tiles_n, g.lower == 1);
    g.tab = nullptr; g.ksplit = 0; g.csplit = 0; g.tri = 0; g.yield = h->cu_yield; g.raise = (h->chain_yield && d.role != 1) ? h->cu_yield : nullptr;
    g.ny = 0; g.ab1 = g.ab2 = g.bb1 = g.bb2 = g.cb1 = g.cb2 = 0;
    const bool plain_k = d.kb0 == 0 && d.kbi == 0 && d.kbj == 0 && d.ke0 < 0 && d.kei == 0 && d.kej == 0;
    if (h->tile_tables && plain_k && !d.rev_m && g.tiles_m < 32768 && g.tiles_n < 32768 && g.ntiles >= 64 && !gemm_takes_small_tiles(h, d)) {
        // equal work per tile: balance the XCDs by tile count (launches with per-tile K ranges keep the formula map)
        const int rc = tile_table(h, g.tiles_m, g.tiles_n, g.lower, g.ls, g.lo, &g.tab, &g.ntiles);
        if (rc) return rc;
    }
    if (g.ntiles == 0) return 0;
    dim3 grid((unsigned)g.ntiles), block(256);
    const bool split = d.split > 1;
    if (split) {
        if (!plain_k || d.rev_m || !d.split_ws) { fvgp_set_error("gemm: split-K needs a plain K range and a workspace"); return -3; }
        const long steps = (d.K / BK + d.split - 1) / d.split;
        g.ksplit = steps * BK; g.csplit = (long)d.M * d.N;
        g.C = d.split_ws; g.ldc = d.N; g.beta = 0.0;
        grid.y = (unsigned)d.split;
        if (d.split_tri) {
            if (d.b_nmajor || d.lower || g.ksplit % 128) { fvgp_set_error("gemm: split_tri needs B (N, K), all tiles, slices of whole 128-blocks"); return -3; }
            g.tri = 1;
        }
    }
    const bool batched = (long)d.batch_y * d.batch_z > 1;
    if (batched) {
        if (split || d.batch_y < 1 || d.batch_z < 1) { fvgp_set_error("gemm: a strided batch excludes split-K"); return -3; }
        g.ny = d.batch_y; g.ab1 = d.a_by; g.ab2 = d.a_bz; g.bb1 = d.b_by; g.bb2 = d.b_bz; g.cb1 = d.c_by; g.cb2 = d.c_bz;
        grid.y = (unsigned)(d.batch_y * d.batch_z);
    }
    if (!split && !batched && gemm_takes_small_tiles(h, d)) {             // too few 128-tiles to fill the chip (the panel chain's steps): 64-tiles
        const long t128 = (long)g.tiles_m * g.tiles_n;
        const dim3 sg((unsigned)(t128 * 4));
        if (!d.b_nmajor) {
            const bool one_stage = d.K == 128 && h->k128_kernels;
            const bool in_place = (const double *)d.C == d.A || (const double *)d.C == d.B;
            const dim3 kg(in_place ? sg.x : (unsigned)(t128 * 8));
            if (in_place) {                                                        // in place: a workgroup owns whole rows
                if (one_stage) hipLaunchKernelGGL((gemm_f64_k128_kernel<128>), kg, block, 0, h->stream, g);
                else hipLaunchKernelGGL((gemm_f64_small_kernel<32, 128, 0>), sg, block, 0, h->stream, g);
            } else {
                if (one_stage) hipLaunchKernelGGL((gemm_f64_k128_kernel<64>), kg, block, 0, h->stream, g);
                else hipLaunchKernelGGL((gemm_f64_small_kernel<64, 64, 0>), sg, block, 0, h->stream, g);
            }
        } else {
            if ((const double *)d.C == d.B)                                        // in place of B (K, N), K == M: whole columns
                hipLaunchKernelGGL((gemm_f64_small_kernel<128, 32, 1>), sg, block, 0, h->stream, g);
            else
                hipLaunchKernelGGL((gemm_f64_small_kernel<64, 64, 1>), sg, block, 0, h->stream, g);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
#define GO(AK, BN) do { if (d.role == 1) hipLaunchKernelGGL((gemm_f64_kernel<AK, BN, 1>), grid, block, 0, h->stream, g); \
                        else hipLaunchKernelGGL((gemm_f64_kernel<AK, BN, 0>), grid, block, 0, h->stream, g); } while (0)
    if (!d.a_kmajor && !d.b_nmajor) GO(0, 0);
    else if (!d.a_kmajor && d.b_nmajor) GO(0, 1);
    else if (d.a_kmajor && !d.b_nmajor) GO(1, 0);
    else GO(1, 1);
#undef GO
    HIPCHK(hipGetLastError());
    if (split) return launch_splitk_reduce(h, d.split_ws, d.split, d.M, d.N, d.lower, d.C, d.ldc, d.beta,
                                           d.split_out ? d.split_out : d.C, d.split_out ? d.split_ldo : d.ldc, g.tri ? g.ksplit : 0);
    return 0;
}

int launch_trsm_tiles(fvgp_handle *h, double *A, int64_t lda, int64_t rows, const double *L, int64_t ldl, const double *dinv) {
    if (rows <= 0) return 0;
    if (rows % 32 || (lda & 1) || (ldl & 1) || ((uintptr_t)A & 15) || ((uintptr_t)L & 15)) { fvgp_set_error("trsm_tiles: rows % 32, even leading dimensions, 16-byte alignment"); return -2; }
    TrsmTilesArgs g{A, (long)lda, L, (long)ldl, dinv, h->chain_yield ? h->cu_yield : nullptr};
    hipLaunchKernelGGL(trsm_tiles_kernel, dim3((unsigned)(rows / 32)), dim3(128), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}
