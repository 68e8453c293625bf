// Vector triangular solves, reductions and small data-movement kernels (HBM-bound work).
//
// potrs with a handful of right-hand sides (y has 1..few columns, fvgp/gp_kv.py:592) is
// bandwidth work: L is streamed exactly once per direction.  Both sweeps use the inverted
// 128x128 diagonal blocks left behind by the factorisation, so a block step is
//   forward :  y_k = inv(L_kk) b_k ;  b[r > k] -= L[r, k-block] y_k      (column panel, rows below)
//   backward:  x_k = inv(L_kk)^T y_k ; y[c < k] -= L[k-block, c]^T x_k   (row panel, columns left)
// Each workgroup recomputes the 128-vector of its step from the L2-resident diagonal
// inverse (128 KiB) instead of waiting on a second launch, so one launch = one block step.
#include "common.h"

namespace {

// wave-wide sum of R independent values per column: all R rows are in flight before any reduction
template <int C, int R>
__device__ __forceinline__ void wave_reduce(double (&acc)[R][C]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) acc[rr][cc] += __shfl_down(acc[rr][cc], off, 64);
}

template <int C>
__global__ __launch_bounds__(256) void fwd_step_kernel(const double *L, long ldl, long np, long k0, const double *linv,
                                                       double *B, long ldb, double *Y, int c_used) {
    __shared__ double sb[128 * C];
    __shared__ double sy[128 * C];
    constexpr int R = 8;                       // rows below in flight per wave and pass
    constexpr int RY = 16;                     // rows of the diagonal inverse in flight per wave and pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // like the backward step, a chain of memory round trips: the first pass over the rows below does not depend on y, so
    // its loads go out before anything else
    const long r0 = k0 + 128;
    const long stride = (long)gridDim.x * 4 * R;
    const long rb0 = r0 + ((long)blockIdx.x * 4 + wave) * R;
    double2_t lf[R];
    if (rb0 < np) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) lf[rr] = *reinterpret_cast<const double2_t *>(L + (rb0 + rr) * ldl + k0 + 2 * lane);
    }
    for (int e = tid; e < 128 * C; e += 256) {
        const int i = e / C, cc = e - i * C;
        sb[e] = cc < c_used ? B[(k0 + i) * ldb + cc] : 0.0;
    }
    __syncthreads();
    // y = Linv * b : wave handles 32 rows, 16 at a time; lanes hold 2 columns each
    {
        double b0[C], b1[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) { b0[cc] = sb[(2 * lane) * C + cc]; b1[cc] = sb[(2 * lane + 1) * C + cc]; }
        for (int i0 = wave * 32; i0 < wave * 32 + 32; i0 += RY) {
            double2_t l2[RY];
#pragma unroll
            for (int rr = 0; rr < RY; ++rr) l2[rr] = *reinterpret_cast<const double2_t *>(linv + (i0 + rr) * 128 + 2 * lane);
            double acc[RY][C];
#pragma unroll
            for (int rr = 0; rr < RY; ++rr)
#pragma unroll
                for (int cc = 0; cc < C; ++cc) acc[rr][cc] = l2[rr][0] * b0[cc] + l2[rr][1] * b1[cc];
            wave_reduce<C, RY>(acc);
            if (lane == 0)
#pragma unroll
                for (int rr = 0; rr < RY; ++rr)
#pragma unroll
                    for (int cc = 0; cc < C; ++cc) sy[(i0 + rr) * C + cc] = acc[rr][cc];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < 128 * C; e += 256) {
            const int i = e / C, cc = e - i * C;
            if (cc < c_used) Y[(k0 + i) * C + cc] = sy[e];
        }
    }
    // rows below: each wave takes R consecutive rows per pass
    double y0[C], y1[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) { y0[cc] = sy[(2 * lane) * C + cc]; y1[cc] = sy[(2 * lane + 1) * C + cc]; }
    for (long rb = rb0; rb < np; rb += stride) {   // np - r0 is a multiple of 128
        double2_t l2[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
            l2[rr] = rb == rb0 ? lf[rr] : *reinterpret_cast<const double2_t *>(L + (rb + rr) * ldl + k0 + 2 * lane);
        double acc[R][C];
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) acc[rr][cc] = l2[rr][0] * y0[cc] + l2[rr][1] * y1[cc];
        wave_reduce<C, R>(acc);
        if (lane == 0)
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int cc = 0; cc < C; ++cc) if (cc < c_used) B[(rb + rr) * ldb + cc] -= acc[rr][cc];
    }
}

// One block step of the backward sweep.  A step is a chain of dependent memory round trips (y_k, the diagonal inverse, the
// row panel), not bandwidth, so each phase has ALL its loads in flight at once: the 128 x 128 inverse as 64 loads per thread,
// the row panel with the 128 rows split over the four waves (32 loads per thread, two columns each) and the four partial
// sums combined through LDS -- three round trips per step instead of thirteen.  Workgroup b owns columns [128 b, 128 b + 128).
template <int C>
__global__ __launch_bounds__(256) void bwd_step_kernel(const double *L, long ldl, long np, long k0, const double *linv,
                                                       double *Yres, double *X, long ldx, int c_used) {
    __shared__ double sy[128 * C];
    __shared__ double sxv[2][128 * C];
    __shared__ double sp[4][128 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the row panel does not depend on x: its loads go out first and fly during the x phase
    const long c2 = (long)blockIdx.x * 128 + 2 * lane;
    const bool have = c2 < k0;
    double2_t l2[32];
    if (have) {
#pragma unroll
        for (int u = 0; u < 32; ++u) l2[u] = *reinterpret_cast<const double2_t *>(L + (k0 + wave * 32 + u) * ldl + c2);
    }
    for (int e = tid; e < 128 * C; e += 256) sy[e] = Yres[k0 * C + e];
    // x = Linv^T y : thread (i, half) walks half of column i of Linv (coalesced across i)
    {
        const int i = tid & 127, half = tid >> 7;
        double l[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) l[u] = linv[(half * 64 + u) * 128 + i];      // Linv[j][i] = 0 for j < i
        __syncthreads();
        double acc[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) acc[cc] = 0.0;
#pragma unroll
        for (int u = 0; u < 64; ++u)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) acc[cc] = fma(l[u], sy[(half * 64 + u) * C + cc], acc[cc]);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) sxv[half][i * C + cc] = acc[cc];
    }
    __syncthreads();
    for (int e = tid; e < 128 * C; e += 256) sxv[0][e] += sxv[1][e];
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < 128 * C; e += 256) {
            const int i = e / C, cc = e - i * C;
            if (cc < c_used) X[(k0 + i) * ldx + cc] = sxv[0][e];
        }
    }
    // columns to the left: this wave's 32 rows of the panel times x, two adjacent columns per lane
    double a0[C], a1[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) { a0[cc] = 0.0; a1[cc] = 0.0; }
    if (have) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const double xv = sxv[0][(wave * 32 + u) * C + cc];
                a0[cc] = fma(l2[u][0], xv, a0[cc]);
                a1[cc] = fma(l2[u][1], xv, a1[cc]);
            }
    }
#pragma unroll
    for (int cc = 0; cc < C; ++cc) { sp[wave][(2 * lane) * C + cc] = a0[cc]; sp[wave][(2 * lane + 1) * C + cc] = a1[cc]; }
    __syncthreads();
    for (int e = tid; e < 128 * C; e += 256) {
        const long col = (long)blockIdx.x * 128 + e / C;
        if (col < k0) Yres[(long)blockIdx.x * 128 * C + e] -= (sp[0][e] + sp[1][e]) + (sp[2][e] + sp[3][e]);
    }
}

// The whole backward sweep in ONE launch (one right-hand side).  The per-block launches above are a chain of N/128 dependent
// kernels of 7-11 us each (x_k by everybody, one read of the row panel, one boundary); here workgroup c owns the 128 columns of
// block c for the whole sweep: it keeps y_c in LDS, walks down the column of L below its block -- the next 128 x 128 block is in
// flight before the x it meets has arrived -- and, once the block right under the diagonal is in, multiplies by the inverse of
// its diagonal block (fetched like one more block of the column) and PUBLISHES x_c to the workgroups left of it.  The hand-off
// is the one form that needs neither fence nor flag (MI355X_MICROARCH.md, inter-workgroup visibility, "granule"): every
// published double travels as ONE naturally aligned 16-byte {value, tag} written by one sc1 (write-through) store and polled by
// sc1 loads until its tag is this launch's -- a stale line shows an old tag and is read again, so nothing depends on when
// another compute unit's caches notice the store.  The tag is a per-handle launch counter, never reused; the stored word is the
// tag XOR a hash of the value, so a granule whose halves come from two different stores does not pass either.  Columns are handed out
// by a ticket in the order the workgroups start (c = nb - 1 - ticket), so a workgroup only ever waits for workgroups that started
// before it: no assumption on the dispatch order, no need for the whole grid to be resident.  Every sum is formed in the order
// the step kernels use (four row quarters of a block, ((0+1)+(2+3)); two halves of the inverse), so the results are
// bit-identical to theirs -- which is how the tests would see a stale read.  Spins are bounded (seconds): a workgroup that gives
// up publishes NaN.
struct BwdSweepArgs {
    const double *L; long ldl; long np;
    const double *linv;           // 128 x 128 inverses of the diagonal blocks, block b at linv + b * 128 * 128 (row-major, zeros above)
    const double *Y;              // np residuals y = L^-1 b
    double *X; long ldx;
    double *gran;                 // np granules of {value, tag} (16 bytes each)
    unsigned long long tag;
    int *ticket;                  // {next column ticket, finished workgroups}; the last workgroup zeroes both
};

__global__ __launch_bounds__(256) void bwd_sweep_kernel(BwdSweepArgs g) {      // 1 workgroup per CU: the next block is in flight beside the current one
    __shared__ double sy[128];
    __shared__ double sx[128];
    __shared__ double sxv[2][128];
    __shared__ double sp[4][128];
    __shared__ int s_c, s_fail;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (int)(g.np / 128);
    if (tid == 0) { s_c = nb - 1 - atomicAdd(g.ticket, 1); s_fail = 0; }
    __syncthreads();
    const int c = s_c;
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(g.gran, 0, 0xffffffff, 0x00020000);
    // (a buffer descriptor per block, the lane's 16 bytes as the only vector offset, the row as a scalar offset: 32 64-bit
    // addresses would cost as many registers as the data)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int rowb = (int)(g.ldl * 8);
    u32x4 l2[32];
    auto load_block = [&](const int I) {
        const double *base = uniform_ptr(g.L + ((long)I * 128 + wave_u * 32) * g.ldl + (long)c * 128);
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int u = 0; u < 32; ++u) l2[u] = __builtin_amdgcn_raw_buffer_load_b128(src, 16 * lane, u * rowb, 0);
    };
    // the inverse of the diagonal block is the last "block" of the column: thread (i, half) takes half of column i of inv(L_cc)
    // (coalesced across i)
    // (registers of its own, requested before anything else: behind the last block of the column its round trip -- 128 KB, 2.6 us -- sat on
    // the path from x_{c+1} to x_c, a third of every hop of the sweep; one workgroup per compute unit has the registers)
    double2_t li[32];
    {
        const double *inv = g.linv + (long)c * 128 * 128 + (tid >> 7) * 64 * 128 + (tid & 127);
#pragma unroll
        for (int u = 0; u < 32; ++u) li[u] = (double2_t){inv[(2 * u) * 128], inv[(2 * u + 1) * 128]};
    }
    if (c < nb - 1) load_block(nb - 1);
    if (tid < 128) sy[tid] = g.Y[(long)c * 128 + tid];
    __syncthreads();
    for (int I = nb - 1; I > c; --I) {
        // ---- x_I: a thread per granule ------------------------------------------------------------------------------------
        if (tid < 128) {
            const int off = (int)(((long)I * 128 + tid) * 16);
            u32x4 v;
            int spins = 0;
            for (;;) {
                v = __builtin_amdgcn_raw_buffer_load_b128(gsrc, off, 0, 16);          // sc1: served past this CU's L1
                // the second word is tag ^ mix(value bits): a granule torn between its 8-byte halves (new tag beside a stale value, or
                // the other way round; never observed on gfx950, not promised by the ISA either) decodes to a wrong tag and is read again
                const unsigned long long vb = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
                const unsigned long long t = ((unsigned long long)v[2] | ((unsigned long long)v[3] << 32)) ^ (vb * 0x9E3779B97F4A7C15ull);
                if (t == g.tag) break;
                if (++spins > (1 << 22)) { s_fail = 1; break; }
                // only the workgroup right behind the frontier is waited for: the further left, the rarer the polls (every
                // poll is a trip past the L1 that the producer's store and the critical poll share the fabric with)
                __builtin_amdgcn_s_sleep(2);
                for (int k = I - 1 - c < 24 ? I - 1 - c : 24; k > 0; --k) __builtin_amdgcn_s_sleep(6);
                asm volatile("" ::: "memory");
            }
            const unsigned long long bits = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
            double xv;
            __builtin_memcpy(&xv, &bits, 8);
            sx[tid] = xv;
        }
        __syncthreads();
        // ---- y_c -= L[I, c]^T x_I: this wave's 32 rows of the block, two adjacent columns per lane ----------------------
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const double xv = sx[wave * 32 + u];
            double2_t lv;
            __builtin_memcpy(&lv, &l2[u], 16);
            a0 = fma(lv[0], xv, a0);
            a1 = fma(lv[1], xv, a1);
        }
        if (I - 1 > c) load_block(I - 1);      // the next block of the column is on its way while the partial sums are combined
        sp[wave][2 * lane] = a0; sp[wave][2 * lane + 1] = a1;
        __syncthreads();
        if (tid < 128) sy[tid] -= (sp[0][tid] + sp[1][tid]) + (sp[2][tid] + sp[3][tid]);
        __syncthreads();
    }
    // ---- x_c = inv(L_cc)^T y_c : thread (i, half) walks half of column i of the inverse ---------------------------------------
    {
        const int i = tid & 127, half = tid >> 7;
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc = fma(li[u][0], sy[half * 64 + 2 * u], acc);
            acc = fma(li[u][1], sy[half * 64 + 2 * u + 1], acc);
        }
        sxv[half][i] = acc;
    }
    __syncthreads();
    if (tid < 128) {
        const double xv = s_fail ? __builtin_nan("") : sxv[0][tid] + sxv[1][tid];
        unsigned long long bits;
        __builtin_memcpy(&bits, &xv, 8);
        const unsigned long long tw = g.tag ^ (bits * 0x9E3779B97F4A7C15ull);        // tag and value vouch for each other (torn granules)
        const u32x4 v = {(unsigned)bits, (unsigned)(bits >> 32), (unsigned)tw, (unsigned)(tw >> 32)};
        __builtin_amdgcn_raw_buffer_store_b128(v, gsrc, (int)(((long)c * 128 + tid) * 16), 0, 16);      // one sc1 store per granule
        g.X[((long)c * 128 + tid) * g.ldx] = xv;
    }
    if (tid == 0 && atomicAdd(g.ticket + 1, 1) == (int)gridDim.x - 1) { atomicExch(g.ticket, 0); atomicExch(g.ticket + 1, 0); }
}

// Sums of 32 per-lane values over the 64 lanes of a wave by halving: at distance 32, 16, 8, 4, 2 a lane keeps one half of its
// list and adds the partner's copy of that half (31 exchanges instead of 32 x 6), distance 1 completes the pair.  Row
// u = 16 b5 + 8 b4 + 4 b3 + 2 b2 + b1 (b_k = bit k of the lane) ends up, complete, in both lanes of its pair.  Fixed order.
__device__ __forceinline__ double wave_sums32(const double (&a)[32][1], const int lane) {
    double v16[16], v8[8], v4[4], v2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double keep = b5 ? a[16 + k][0] : a[k][0], give = b5 ? a[k][0] : a[16 + k][0]; v16[k] = keep + __shfl_xor(give, 32, 64); }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double keep = b4 ? v16[8 + k] : v16[k], give = b4 ? v16[k] : v16[8 + k]; v8[k] = keep + __shfl_xor(give, 16, 64); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double keep = b3 ? v8[4 + k] : v8[k], give = b3 ? v8[k] : v8[4 + k]; v4[k] = keep + __shfl_xor(give, 8, 64); }
#pragma unroll
    for (int k = 0; k < 2; ++k) { const double keep = b2 ? v4[2 + k] : v4[k], give = b2 ? v4[k] : v4[2 + k]; v2[k] = keep + __shfl_xor(give, 4, 64); }
    const double keep = b1 ? v2[1] : v2[0], give = b1 ? v2[0] : v2[1];
    const double v1 = keep + __shfl_xor(give, 2, 64);
    return v1 + __shfl_xor(v1, 1, 64);
}

// The forward sweep L y = b in ONE launch (one right-hand side): the mirror image of bwd_sweep_kernel.  Workgroup r owns block ROW r
// of L -- 128 contiguous rows, streamed left to right, the next 128 x 128 block in flight before the y it meets has arrived -- and
// keeps, per thread, the partial sums of its wave's 32 rows over the two columns its lane reads; they are reduced across the
// lanes once, when the row is through (every sum in a fixed order: nothing depends on timing).  Then t = b_r - sum, y_r =
// inv(L_rr) t with the inverse fetched like one more block of the row, and y_r is PUBLISHED to the workgroups right of it as
// 16-byte {value, tag ^ hash(value)} granules exactly as the backward sweep hands its x to the left.  Rows are handed out by a
// ticket in start order (r = ticket): a workgroup only waits for workgroups that started before it.  Replaces N/128 dependent
// launches of 8-20 us each (the path of a size that leaves no padding row for the fused solve, n % 128 == 0: N = 4096 2.91 ms
// against 2.46 at N = 4000, same factorisation).
struct FwdSweepArgs {
    const double *L; long ldl; long np;
    const double *linv;
    const double *B; long ldb;    // right-hand side, column 0 (np rows, padding rows zero)
    double *Y;                    // np doubles: y = L^-1 b
    double *gran;
    unsigned long long tag;
    int *ticket;
};

__global__ __launch_bounds__(256) void fwd_sweep_kernel(FwdSweepArgs g) {
    __shared__ double sy[2][128];
    __shared__ double ssum[128];
    __shared__ double st[128];
    __shared__ int s_r, s_fail;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_r = atomicAdd(g.ticket, 1); s_fail = 0; }
    __syncthreads();
    const int r = s_r;
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(g.gran, 0, 0xffffffff, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    u32x4 l2[32];
    // this wave's 32 rows of a 128 x 128 block (row stride `rowb` bytes): lane l takes columns 2 l, 2 l + 1 of each
    auto load_rows = [&](const double *blk, const long stride) {
        const double *base = uniform_ptr(blk + (long)wave_u * 32 * stride);
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0xffffffff, 0x00020000);
        const int rowb = (int)(stride * 8);
#pragma unroll
        for (int u = 0; u < 32; ++u) l2[u] = __builtin_amdgcn_raw_buffer_load_b128(src, 16 * lane, u * rowb, 0);
    };
    const double *Lrow = g.L + (long)r * 128 * g.ldl;
    const double *inv = g.linv + (long)r * 128 * 128;
    // (the inverse of the diagonal block in registers of its own, requested first: see bwd_sweep_kernel)
    u32x4 li[32];
    {
        const double *base = uniform_ptr(inv + (long)wave_u * 32 * 128);
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int u = 0; u < 32; ++u) li[u] = __builtin_amdgcn_raw_buffer_load_b128(src, 16 * lane, u * 1024, 0);
    }
    if (r > 0) load_rows(Lrow, g.ldl);
    const double bval = tid < 128 ? g.B[((long)r * 128 + tid) * g.ldb] : 0.0;
    double acc[32][1];
#pragma unroll
    for (int u = 0; u < 32; ++u) acc[u][0] = 0.0;
    for (int c = 0; c < r; ++c) {
        if (tid < 128) {
            const int off = (int)(((long)c * 128 + tid) * 16);
            u32x4 v;
            int spins = 0;
            for (;;) {
                v = __builtin_amdgcn_raw_buffer_load_b128(gsrc, off, 0, 16);          // sc1: served past this CU's L1
                const unsigned long long vb = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
                const unsigned long long t = ((unsigned long long)v[2] | ((unsigned long long)v[3] << 32)) ^ (vb * 0x9E3779B97F4A7C15ull);
                if (t == g.tag) break;
                if (++spins > (1 << 22)) { s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(2);
                for (int k = r - 1 - c < 24 ? r - 1 - c : 24; k > 0; --k) __builtin_amdgcn_s_sleep(6);      // far behind the frontier: rare polls
                asm volatile("" ::: "memory");
            }
            const unsigned long long bits = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
            double yv;
            __builtin_memcpy(&yv, &bits, 8);
            sy[c & 1][tid] = yv;
        }
        __syncthreads();
        const double y0 = sy[c & 1][2 * lane], y1 = sy[c & 1][2 * lane + 1];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            double2_t lv;
            __builtin_memcpy(&lv, &l2[u], 16);
            acc[u][0] = fma(lv[0], y0, acc[u][0]);
            acc[u][0] = fma(lv[1], y1, acc[u][0]);
        }
        if (c + 1 < r) load_rows(Lrow + (long)(c + 1) * 128, g.ldl);      // on its way while the next y is awaited
    }
    const int urow = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    {
        const double tot = wave_sums32(acc, lane);
        if (!(lane & 1)) ssum[wave * 32 + urow] = tot;
    }
    __syncthreads();
    if (tid < 128) st[tid] = bval - ssum[tid];
    __syncthreads();
    {
        const double t0 = st[2 * lane], t1 = st[2 * lane + 1];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            double2_t lv;
            __builtin_memcpy(&lv, &li[u], 16);
            acc[u][0] = fma(lv[1], t1, lv[0] * t0);
        }
    }
    {
        const double tot = wave_sums32(acc, lane);
        if (!(lane & 1)) ssum[wave * 32 + urow] = tot;
    }
    __syncthreads();
    if (tid < 128) {
        const double yv = s_fail ? __builtin_nan("") : ssum[tid];
        unsigned long long bits;
        __builtin_memcpy(&bits, &yv, 8);
        const unsigned long long tw = g.tag ^ (bits * 0x9E3779B97F4A7C15ull);
        const u32x4 v = {(unsigned)bits, (unsigned)(bits >> 32), (unsigned)tw, (unsigned)(tw >> 32)};
        __builtin_amdgcn_raw_buffer_store_b128(v, gsrc, (int)(((long)r * 128 + tid) * 16), 0, 16);      // one sc1 store per granule
        g.Y[(long)r * 128 + tid] = yv;
    }
    if (tid == 0 && atomicAdd(g.ticket + 1, 1) == (int)gridDim.x - 1) { atomicExch(g.ticket, 0); atomicExch(g.ticket + 1, 0); }
}

__global__ void diag_logsum_kernel(const double *L, long n, long ldl, double *out) {
    __shared__ double sw[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += log(fabs(L[i * ldl + i]));
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[0] = 2.0 * t; }
}

__global__ void sum_kernel(const double *v, long n, double *out) {
    __shared__ double sw[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[0] = t; }
}

// out[0] = -sum log v[i]  (v = the reciprocal diagonal the leaves leave: sum log L_ii), fixed order
__global__ void neg_log_sum_kernel(const double *v, long n, double *out) {
    __shared__ double sw[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s -= log(v[i]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[0] = t; }
}

// out[0] = sum_{i<n, cc<c} a[i*lda+cc] * b[i*ldb+cc]
__global__ void dot_rows_kernel(const double *a, long lda, const double *b, long ldb, long n, int c, double *out) {
    __shared__ double sw[16];
    double s = 0.0;
    const long tot = n * c;
    for (long e = threadIdx.x; e < tot; e += blockDim.x) { const long i = e / c; const int cc = (int)(e - i * c); s = fma(a[i * lda + cc], b[i * ldb + cc], s); }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[0] = t; }
}

// dst (rows_pad, ldd) <- src (rows, lds) for [rows x cols], zero elsewhere up to rows_pad x cols_pad
__global__ void copy_cols_kernel(const double *src, long lds, double *dst, long ldd, long rows, long cols, long rows_pad, long cols_pad) {
    const long tot = rows_pad * cols_pad;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long i = e / cols_pad, j = e - i * cols_pad;
        dst[i * ldd + j] = (i < rows && j < cols) ? src[i * lds + j] : 0.0;
    }
}

__global__ void symmetrize_kernel(double *A, long n, long lda) {
    // 32x32 tile transpose through LDS; only tiles with bi > bj (and the diagonal tiles in place)
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int rr = ty; rr < 32; rr += 8) {
        const long i = (long)bi * 32 + rr, j = (long)bj * 32 + tx;
        t[rr][tx] = (i < n && j < n) ? A[i * lda + j] : 0.0;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        const long i = (long)bj * 32 + rr, j = (long)bi * 32 + tx;   // target (i,j) in the upper triangle = source (j,i)
        if (i < n && j < n && j > i) A[i * lda + j] = t[tx][rr];
    }
}

// out[p] = base - sum_{i<rows} V[i*ldv+p]^2, p < P   (posterior variance: k(x_p,x_p) - |L^-1 k_p|^2)
__global__ void colsumsq_kernel(const double *V, long rows, long ldv, long P, double base, double *out, double sign) {
    __shared__ double sp[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long p = (long)blockIdx.x * 32 + tx;
    double s = 0.0;
    if (p < P) for (long i = ty; i < rows; i += 8) { const double v = V[i * ldv + p]; s = fma(v, v, s); }
    sp[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && p < P) { double t = 0.0; for (int k = 0; k < 8; ++k) t += sp[k][tx]; out[p] = base - sign * t; }
}

// out[p] = sum_{i<rows} A[i*lda+p] * B[i*ldb+p], p < P   (column-wise dot products: k^T (KV^-1 k) per prediction point)
__global__ void coldot_kernel(const double *A, long lda, const double *B, long ldb, long rows, long P, double *out) {
    __shared__ double sp[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long p = (long)blockIdx.x * 32 + tx;
    double s = 0.0;
    if (p < P) for (long i = ty; i < rows; i += 8) s = fma(A[i * lda + p], B[i * ldb + p], s);
    sp[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && p < P) { double t = 0.0; for (int k = 0; k < 8; ++k) t += sp[k][tx]; out[p] = t; }
}

// row-wise twins of the two kernels above for the transposed cross-covariance block KT (P x n): one workgroup per
// prediction point streams its row (16-byte loads, coalesced), block sums in a fixed order
template <int C>
__global__ __launch_bounds__(256) void rows_dot_kernel(const double *KT, long ldk, const double *alpha, long lda, int ncol, long n,
                                                       double *out, long ldo) {
    __shared__ double sp[4][C];
    const long p = blockIdx.x;
    const double *row = KT + p * ldk;
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0;
    for (long i = 2L * threadIdx.x; i < n; i += 512) {        // n is read up to the even index below it; the odd tail follows
        if (i + 1 < n) {
            const double2_t kv = *reinterpret_cast<const double2_t *>(row + i);
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (c < ncol) acc[c] = fma(kv[1], alpha[(i + 1) * lda + c], fma(kv[0], alpha[i * lda + c], acc[c]));
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) if (c < ncol) acc[c] = fma(row[i], alpha[i * lda + c], acc[c]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += __shfl_down(acc[c], off, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < C; ++c) sp[threadIdx.x >> 6][c] = acc[c];
    __syncthreads();
    if (threadIdx.x < ncol) out[p * ldo + threadIdx.x] = (sp[0][threadIdx.x] + sp[1][threadIdx.x]) + (sp[2][threadIdx.x] + sp[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void rows_sumsq_base_kernel(const double *KT, long ldk, long n, double base, double *out) {
    __shared__ double sp[4];
    const long p = blockIdx.x;
    const double *row = KT + p * ldk;
    double s = 0.0;
    for (long i = 2L * threadIdx.x; i < n; i += 512) {
        if (i + 1 < n) { const double2_t v = *reinterpret_cast<const double2_t *>(row + i); s = fma(v[1], v[1], fma(v[0], v[0], s)); }
        else s = fma(row[i], row[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sp[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[p] = base - ((sp[0] + sp[1]) + (sp[2] + sp[3]));
}

// C = beta C + sum over the splits of the partial products (alpha already applied), splits added in index order
// tri_ks > 0 (GemmDesc::split_tri): tile column tj only has the slices that start below K = 128 (tj + 1)
__global__ void splitk_reduce_kernel(const double *ws, int split, long M, long N, int lower, const double *C, long ldc, double beta,
                                     double *out, long ldo, long tri_ks) {
    const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (e >= M * N) return;
    const long i = e / N, j = e - i * N;
    if (lower && (j >> 7) > (i >> 7)) return;
    if (tri_ks) { const long nz = (128 * ((j >> 7) + 1) + tri_ks - 1) / tri_ks; if (nz < split) split = (int)nz; }
    double2_t s = *reinterpret_cast<const double2_t *>(ws + e);
    for (int z = 1; z < split; ++z) { const double2_t t = *reinterpret_cast<const double2_t *>(ws + (long)z * M * N + e); s[0] += t[0]; s[1] += t[1]; }
    if (beta != 0.0) {
        const double2_t o = *reinterpret_cast<const double2_t *>(C + i * ldc + j);
        s[0] = fma(beta, o[0], s[0]); s[1] = fma(beta, o[1], s[1]);
    }
    *reinterpret_cast<double2_t *>(out + i * ldo + j) = s;
}

// seed of the 1024-block inverses: block row b (128 x 1024) of W <- zeros, with inv(L_bb) at its place on the diagonal of
// the 1024-block the row belongs to
__global__ __launch_bounds__(256) void winv_seed_kernel(const double *linv, double *W, int wshift) {
    const long b = blockIdx.x;
    const int w = 1 << wshift;                       // width of the blocks being inverted (1024; 512 ... 2048 for a panel's square)
    const int c0 = (int)(b & ((w >> 7) - 1)) * 128;
    const double *src = linv + b * 128 * 128;
    double *dst = W + b * 128 * w;
    // blockIdx.y: sixteen rows of the block row each (eight workgroups per 128 rows: a panel's seed is on the chain's path)
    for (int e = (blockIdx.y * 16 << wshift) + threadIdx.x * 2; e < ((blockIdx.y + 1) * 16 << wshift); e += 512) {
        const int i = e >> wshift, j = e & (w - 1);
        double2_t v = {0.0, 0.0};
        if (j >= c0 && j < c0 + 128) v = *reinterpret_cast<const double2_t *>(src + i * 128 + (j - c0));
        *reinterpret_cast<double2_t *>(dst + e) = v;
    }
}

// copy the 128x128 tiles on and below the block diagonal (np a multiple of 128)
__global__ void copy_lower_tiles_kernel(const double *src, long lds, double *dst, long ldd) {
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (tj > ti) return;
    for (int e = threadIdx.x; e < 128 * 64; e += blockDim.x) {
        const int rr = e >> 6, c2 = (e & 63) * 2;
        const long off_s = ((long)ti * 128 + rr) * lds + (long)tj * 128 + c2;
        const long off_d = ((long)ti * 128 + rr) * ldd + (long)tj * 128 + c2;
        *reinterpret_cast<double2_t *>(dst + off_d) = *reinterpret_cast<const double2_t *>(src + off_s);
    }
}

// dst (cols x rows) <- src^T for a rows x cols matrix, both multiples of 32, in 32 x 32 pieces through LDS
__global__ void transpose_kernel(const double *src, long lds, double *dst, long ldd) {
    __shared__ double t[32][33];
    const long bj = blockIdx.x, bi = blockIdx.y;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int rr = ty; rr < 32; rr += 8) t[rr][tx] = src[(bi * 32 + rr) * lds + bj * 32 + tx];
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) dst[(bj * 32 + rr) * ldd + bi * 32 + tx] = t[tx][rr];
}

// dst tile (tj, ti) <- transpose of src tile (ti, tj) for the 128 x 128 tiles on and below the block diagonal, in 32 x 32
// pieces through LDS (the upper part of src's diagonal tiles is explicit zeros, so dst's diagonal tiles come out with a zero
// lower part): W = inv(L) (lower, k-major for W^T W) becomes W^T (upper, k-minor: the (M,K) x (N,K) layout of the fast GEMM)
// `mirror`: in place (dst == src), only the pieces strictly below the diagonal -- a symmetric matrix whose 128-tiles on and below
// the block diagonal were computed gets its upper part (the diagonal tiles were computed whole)
__global__ void transpose_lower_tiles_kernel(const double *src, long lds, double *dst, long ldd, int mirror) {
    __shared__ double t[32][33];
    const int bj = blockIdx.x, bi = blockIdx.y;           // 32-granular block coordinates in src
    if ((bj >> 2) > (bi >> 2)) return;
    if (mirror && bj >= bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int rr = ty; rr < 32; rr += 8) t[rr][tx] = src[((long)bi * 32 + rr) * lds + (long)bj * 32 + tx];
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) dst[((long)bj * 32 + rr) * ldd + (long)bi * 32 + tx] = t[tx][rr];
}

// partial[block] = sum over the block's rows i and all j < n of (W[i][j] - b_i b_j) D[i][j]: the trace term of the gradient
// for a derivative matrix D that exists as numbers (host kernel callables, matrix-valued noise derivatives),
// tr(KV^-1 dK) - b^T dK b with W = KV^-1 symmetric and full (gp_marginal_likelihood.py:301-306); b may be null
__global__ __launch_bounds__(256) void trace_dot_kernel(const double *W, long ldw, const double *D, long ldd, const double *b, long ldb,
                                                        long n, double *partial) {
    __shared__ double sw[4];
    double s = 0.0;
    for (long i = blockIdx.x; i < n; i += gridDim.x) {
        const double bi = b ? b[i * ldb] : 0.0;
        for (long j = threadIdx.x; j < n; j += 256) {
            const double bj = b ? b[j * ldb] : 0.0;
            s = fma(W[i * ldw + j] - bi * bj, D[i * ldd + j], s);
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// A[i][j] += alpha * B[i][j] on the lower triangle (j <= i < n): K + V for a matrix-valued noise model (gp_kv.py:654-657)
__global__ void add_lower_kernel(double *A, long lda, const double *B, long ldb, long n, double alpha) {
    for (long i = blockIdx.y; i < n; i += gridDim.y)
        for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j <= i; j += (long)gridDim.x * blockDim.x)
            A[i * lda + j] = fma(alpha, B[i * ldb + j], A[i * lda + j]);
}

// A[i][j] += alpha * B[i][j] on a rows x cols rectangle
__global__ void add_matrix_kernel(double *A, long lda, const double *B, long ldb, long rows, long cols, double alpha) {
    const long tot = rows * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long i = e / cols, j = e - i * cols;
        A[i * lda + j] = fma(alpha, B[i * ldb + j], A[i * lda + j]);
    }
}

// rows n..np-1 of a padded square matrix <- identity rows (lower part; the strict upper is never read)
__global__ void pad_identity_kernel(double *A, long n, long np, long lda) {
    const long rows = np - n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < rows * np; e += (long)gridDim.x * blockDim.x) {
        const long i = n + e / np, j = e % np;
        if (j <= i) A[i * lda + j] = (i == j) ? 1.0 : 0.0;
    }
}

// rows n..n+ncol-1 of the padded K+V <- (y-m)^T with a diagonal entry 1 + sum|y-m|^2 / min(V), which
// dominates |L^-1 (y-m)|^2 <= |y-m|^2 / lambda_min(K+V) and so keeps the appended block positive definite
__global__ void rhs_rows_kernel(double *A, long n, long lda, const double *ymean, int ncol, const double *vdiag) {
    __shared__ double ssum[16], smin[16];
    __shared__ double sbig;
    double s = 0.0, mn = 1e300;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        for (int c = 0; c < ncol; ++c) { const double v = ymean[i * ncol + c]; s = fma(v, v, s); }
        const double vv = vdiag[i]; mn = vv < mn ? vv : mn;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); const double o = __shfl_down(mn, off, 64); mn = o < mn ? o : mn; }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; smin[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, m2 = 1e300;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { t += ssum[w]; m2 = smin[w] < m2 ? smin[w] : m2; }
        sbig = 1.0 + t / m2;
    }
    __syncthreads();
    const double big = sbig;
    for (long e = threadIdx.x; e < (long)ncol * (n + ncol); e += blockDim.x) {
        const int c = (int)(e / (n + ncol)); const long j = e % (n + ncol);
        double v;
        if (j < n) v = ymean[j * ncol + c];
        else v = (j - n == c) ? big : 0.0;
        if (j <= n + c) A[(n + c) * lda + j] = v;
    }
}

// out[0] = sum over rows row0..row0+nrows-1, cols 0..ncols-1 of A^2
__global__ void rowsumsq_kernel(const double *A, long lda, long row0, int nrows, long ncols, double *out) {
    __shared__ double sw[16];
    double s = 0.0;
    for (int rr = 0; rr < nrows; ++rr)
        for (long j = threadIdx.x; j < ncols; j += blockDim.x) { const double v = A[(row0 + rr) * lda + j]; s = fma(v, v, s); }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[0] = t; }
}

// vec (np x C) <- transpose of rows row0.. of A (first row0 columns), zero elsewhere
__global__ void rows_to_vec_kernel(const double *A, long lda, long row0, int nrows, double *vec, int C, long np) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < np * C; e += (long)gridDim.x * blockDim.x) {
        const long i = e / C; const int c = (int)(e % C);
        vec[e] = (i < row0 && c < nrows) ? A[(row0 + c) * lda + i] : 0.0;
    }
}

// What follows the fused factorisation of fvgp_hip_loglik in ONE launch (four launches of ~5 us each at the training loop's sizes):
// block 0: out[0] = -sum log v[i] (neg_log_sum_kernel); block 1: out[1] = the squared norm of the appended rows (rowsumsq_kernel);
// the other blocks: vec <- the appended rows transposed (rows_to_vec_kernel), alpha <- 0.  The two sums add up in the order of the
// kernels they replace (1024 threads striding the input, wave shuffle, sixteen partial sums in turn).
__global__ __launch_bounds__(1024) void loglik_tail_kernel(const double *v, long nlog, const double *A, long lda, long row0, int nrows, double *out,
                                                         double *vec, int C, long np, double *alpha, int ncol) {
    __shared__ double sw[16];
    if (blockIdx.x <= 1) {
        double s = 0.0;
        if (blockIdx.x == 0) { for (long i = threadIdx.x; i < nlog; i += blockDim.x) s -= log(v[i]); }
        else {
            for (int rr = 0; rr < nrows; ++rr)
                for (long j = threadIdx.x; j < row0; j += blockDim.x) { const double a = A[(row0 + rr) * lda + j]; s = fma(a, a, s); }
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sw[w]; out[blockIdx.x] = t; }
        return;
    }
    if (!vec) return;
    const long stride = (long)(gridDim.x - 2) * blockDim.x, first = (long)(blockIdx.x - 2) * blockDim.x + threadIdx.x;
    for (long e = first; e < np * C; e += stride) {
        const long i = e / C; const int c = (int)(e % C);
        vec[e] = (i < row0 && c < nrows) ? A[(row0 + c) * lda + i] : 0.0;
    }
    for (long e = first; e < np * ncol; e += stride) alpha[e] = 0.0;
}

}  // namespace

int launch_loglik_tail(fvgp_handle *h, const double *v, int64_t nlog, const double *A, int64_t lda, int64_t n, int ncol, double *out2_dev,
                       double *vec, int C, int64_t np, double *alpha) {
    long blocks = vec ? (np * C + 1023) / 1024 : 0; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(loglik_tail_kernel, dim3((unsigned)(2 + blocks)), dim3(1024), 0, h->stream, v, (long)nlog, A, (long)lda, (long)n, ncol, out2_dev,
                       vec, C, (long)np, alpha, ncol);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_rhs_rows(fvgp_handle *h, double *A, int64_t n, int64_t lda, const double *ymean, int ncol, const double *vdiag) {
    hipLaunchKernelGGL(rhs_rows_kernel, dim3(1), dim3(1024), 0, h->stream, A, (long)n, (long)lda, ymean, ncol, vdiag);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_rowsumsq(fvgp_handle *h, const double *A, int64_t lda, int64_t row0, int nrows, int64_t ncols, double *out_dev) {
    hipLaunchKernelGGL(rowsumsq_kernel, dim3(1), dim3(1024), 0, h->stream, A, (long)lda, (long)row0, nrows, (long)ncols, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_rows_to_vec(fvgp_handle *h, const double *A, int64_t lda, int64_t row0, int nrows, double *vec, int C, int64_t np) {
    long tot = np * C; long blocks = (tot + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(rows_to_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, A, (long)lda, (long)row0, nrows, vec, C, (long)np);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_add_lower(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t n, double alpha) {
    long bx = (n + 255) / 256; if (bx > 64) bx = 64;
    const long by = n < 65535 ? n : 65535;                 // gridDim.y is limited to 65535: the rows are walked with that stride
    hipLaunchKernelGGL(add_lower_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, h->stream, A, (long)lda, B, (long)ldb, (long)n, alpha);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_add_matrix(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double alpha) {
    long tot = rows * cols;
    if (tot <= 0) return 0;
    long blocks = (tot + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_matrix_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, A, (long)lda, B, (long)ldb, (long)rows, (long)cols, alpha);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_pad_identity(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda) {
    if (np <= n) return 0;
    long tot = (np - n) * np;
    long blocks = (tot + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, A, (long)n, (long)np, (long)lda);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int C>
static int fwd_go(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv, double *B, int64_t ldb, double *Y, int c) {
    long rows = np - k0 - 128;
    long blocks = rows > 0 ? (rows + 31) / 32 : 1;   // 8 rows per wave-pass, 4 waves
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((fwd_step_kernel<C>), dim3((unsigned)blocks), dim3(256), 0, h->stream, L, (long)ldl, (long)np, (long)k0, linv, B, (long)ldb, Y, c);
    return 0;
}

int launch_fwd_step(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv_k,
                    double *B, int64_t ldb, double *Y, int c) {
    if (c <= 1) fwd_go<1>(h, L, ldl, np, k0, linv_k, B, ldb, Y, c);
    else if (c <= 2) fwd_go<2>(h, L, ldl, np, k0, linv_k, B, ldb, Y, c);
    else if (c <= 4) fwd_go<4>(h, L, ldl, np, k0, linv_k, B, ldb, Y, c);
    else fwd_go<8>(h, L, ldl, np, k0, linv_k, B, ldb, Y, c);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int C>
static int bwd_go(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv, double *Yres, double *X, int64_t ldx, int c) {
    long blocks = k0 > 0 ? k0 / 128 : 1;             // one workgroup per 128 columns to the left (k0 is a multiple of 128)
    hipLaunchKernelGGL((bwd_step_kernel<C>), dim3((unsigned)blocks), dim3(256), 0, h->stream, L, (long)ldl, (long)np, (long)k0, linv, Yres, X, (long)ldx, c);
    return 0;
}

int launch_bwd_step(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv_k,
                    double *Yres, double *X, int64_t ldx, int c) {
    if (c <= 1) bwd_go<1>(h, L, ldl, np, k0, linv_k, Yres, X, ldx, c);
    else if (c <= 2) bwd_go<2>(h, L, ldl, np, k0, linv_k, Yres, X, ldx, c);
    else if (c <= 4) bwd_go<4>(h, L, ldl, np, k0, linv_k, Yres, X, ldx, c);
    else bwd_go<8>(h, L, ldl, np, k0, linv_k, Yres, X, ldx, c);
    HIPCHK(hipGetLastError());
    return 0;
}

// granules (16 bytes per row, tags of earlier launches never match) and the ticket words of the one-launch sweeps
static int sweep_buffers(fvgp_handle *h, int64_t np) {
    const size_t need = (size_t)np * 2;                              // doubles: 16 bytes per granule
    if (need > h->sweep_gran_cap) {
        if (h->sweep_gran) HIPCHK(hipFree(h->sweep_gran));
        h->sweep_gran = nullptr; h->sweep_gran_cap = 0;
        HIPCHK(hipMalloc((void **)&h->sweep_gran, need * sizeof(double)));
        HIPCHK(hipMemset(h->sweep_gran, 0, need * sizeof(double)));
        h->sweep_gran_cap = need;
    }
    if (!h->sweep_ticket) {
        HIPCHK(hipMalloc((void **)&h->sweep_ticket, 2 * sizeof(int)));
        HIPCHK(hipMemset(h->sweep_ticket, 0, 2 * sizeof(int)));
    }
    return 0;
}

// the whole backward sweep in one launch (bwd_sweep_kernel, one right-hand side); Yres is only read
int launch_bwd_sweep(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, const double *linv, const double *Yres, double *X, int64_t ldx, int c) {
    if (c != 1) { fvgp_set_error("bwd_sweep: one right-hand side"); return -9; }
    int rc = sweep_buffers(h, np); if (rc) return rc;
    BwdSweepArgs g{L, (long)ldl, (long)np, linv, Yres, X, (long)ldx, h->sweep_gran, ++h->sweep_tag, h->sweep_ticket};
    hipLaunchKernelGGL(bwd_sweep_kernel, dim3((unsigned)(np / 128)), dim3(256), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

// the whole forward sweep in one launch (fwd_sweep_kernel, one right-hand side): Y (np doubles) <- L^-1 B[:, 0]; B is only read
int launch_fwd_sweep(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, const double *linv, const double *B, int64_t ldb, double *Y) {
    int rc = sweep_buffers(h, np); if (rc) return rc;
    FwdSweepArgs g{L, (long)ldl, (long)np, linv, B, (long)ldb, Y, h->sweep_gran, ++h->sweep_tag, h->sweep_ticket};
    hipLaunchKernelGGL(fwd_sweep_kernel, dim3((unsigned)(np / 128)), dim3(256), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_diag_logsum(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *out_dev) {
    hipLaunchKernelGGL(diag_logsum_kernel, dim3(1), dim3(1024), 0, h->stream, L, (long)n, (long)ldl, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_sum(fvgp_handle *h, const double *v, int64_t n, double *out_dev) {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, h->stream, v, (long)n, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_neg_log_sum(fvgp_handle *h, const double *v, int64_t n, double *out_dev) {
    hipLaunchKernelGGL(neg_log_sum_kernel, dim3(1), dim3(1024), 0, h->stream, v, (long)n, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_dot_rows(fvgp_handle *h, const double *a, int64_t lda, const double *b, int64_t ldb, int64_t n, int c, double *out_dev) {
    hipLaunchKernelGGL(dot_rows_kernel, dim3(1), dim3(1024), 0, h->stream, a, (long)lda, b, (long)ldb, (long)n, c, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_copy_cols(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t rows, int64_t cols,
                     int64_t rows_pad, int64_t cols_pad) {
    long tot = rows_pad * cols_pad;
    if (tot <= 0) return 0;
    long blocks = (tot + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, src, (long)lds, dst, (long)ldd, (long)rows, (long)cols, (long)rows_pad, (long)cols_pad);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_symmetrize(fvgp_handle *h, double *A, int64_t n, int64_t lda) {
    unsigned nb = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, h->stream, A, (long)n, (long)lda);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_colsumsq(fvgp_handle *h, const double *V, int64_t rows, int64_t ldv, int64_t P, double base, double *out, double sign) {
    hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, h->stream, V, (long)rows, (long)ldv, (long)P, base, out, sign);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_coldot(fvgp_handle *h, const double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t P, double *out) {
    hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, h->stream, A, (long)lda, B, (long)ldb, (long)rows, (long)P, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_rows_dot(fvgp_handle *h, const double *KT, int64_t ldk, const double *alpha, int64_t lda, int ncol, int64_t n, int64_t P,
                    double *out, int64_t ldo) {
    if (ncol < 1 || ncol > 8) return -5;
    if ((ldk & 1) || ((uintptr_t)KT & 15)) return -2;
    const int C = ncol <= 1 ? 1 : ncol <= 2 ? 2 : ncol <= 4 ? 4 : 8;
    dim3 grid((unsigned)P), block(256);
#define GO(CC) hipLaunchKernelGGL((rows_dot_kernel<CC>), grid, block, 0, h->stream, KT, (long)ldk, alpha, (long)lda, ncol, (long)n, out, (long)ldo)
    if (C == 1) GO(1); else if (C == 2) GO(2); else if (C == 4) GO(4); else GO(8);
#undef GO
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_rows_sumsq_base(fvgp_handle *h, const double *KT, int64_t ldk, int64_t n, int64_t P, double base, double *out) {
    if ((ldk & 1) || ((uintptr_t)KT & 15)) return -2;
    hipLaunchKernelGGL(rows_sumsq_base_kernel, dim3((unsigned)P), dim3(256), 0, h->stream, KT, (long)ldk, (long)n, base, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_splitk_reduce(fvgp_handle *h, const double *ws, int split, int64_t M, int64_t N, int lower, const double *C, int64_t ldc, double beta,
                         double *out, int64_t ldo, int64_t tri_ksplit) {
    const long pairs = (long)M * N / 2;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, ws, split, (long)M, (long)N, lower,
                       C, (long)ldc, beta, out, (long)ldo, (long)tri_ksplit);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_winv_seed(fvgp_handle *h, const double *linv, int64_t nblk, double *W, int64_t w) {
    if (nblk <= 0) return 0;
    int wshift = 7;
    while ((1L << wshift) < w) ++wshift;
    if ((1L << wshift) != w || w > 8192) { fvgp_set_error("block inverses: the width must be 128 times a power of two"); return -5; }
    hipLaunchKernelGGL(winv_seed_kernel, dim3((unsigned)nblk, 8), dim3(256), 0, h->stream, linv, W, wshift);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_transpose(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    if (rows % 32 || cols % 32) { fvgp_set_error("transpose: multiples of 32"); return -5; }
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(cols / 32), (unsigned)(rows / 32)), dim3(256), 0, h->stream, src, (long)lds, dst, (long)ldd);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_transpose_lower_tiles(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t np) {
    unsigned nb = (unsigned)(np / 32);
    if (nb == 0) return 0;
    hipLaunchKernelGGL(transpose_lower_tiles_kernel, dim3(nb, nb), dim3(256), 0, h->stream, src, (long)lds, dst, (long)ldd, src == dst ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_trace_dot(fvgp_handle *h, const double *W, int64_t ldw, const double *D, int64_t ldd, const double *b, int64_t ldb, int64_t n,
                     double *partial, int *nblocks) {
    const long blocks = n < 2048 ? n : 2048;
    hipLaunchKernelGGL(trace_dot_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, W, (long)ldw, D, (long)ldd, b, (long)ldb, (long)n, partial);
    HIPCHK(hipGetLastError());
    *nblocks = (int)blocks;
    return 0;
}

int launch_copy_lower_tiles(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t np) {
    unsigned nb = (unsigned)(np / 128);
    if (nb == 0) return 0;
    hipLaunchKernelGGL(copy_lower_tiles_kernel, dim3(nb, nb), dim3(256), 0, h->stream, src, (long)lds, dst, (long)ldd);
    HIPCHK(hipGetLastError());
    return 0;
}
