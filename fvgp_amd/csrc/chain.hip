// The panel chain as ONE resident kernel per panel.
//
// A panel of the blocked Cholesky (scipy.linalg.cho_factor -> dpotrf, fvgp/gp_lin_alg.py:245) is w = 128 n columns wide and
// reaches from its diagonal square down to the last row of the matrix.  Per 128 columns the chain is: factor the diagonal
// block (leaf), solve every block below against it, bring the rest of the panel up to date.  Here the whole panel is one
// launch whose workgroups hand results to each other through memory -- a workgroup per 128 x 128 BLOCK of the panel:
//
//   diagonal block (k, k)   keeps its block in registers from the start (36 lower 16 x 16 tiles over eight waves), subtracts
//                           X X^T for every solved block X = L[k, j] of its row, one 16-column tile column at a time BEHIND the
//                           solver's column flags (syrk_follow), then runs the block's LEAF (leaf_body.h) straight from LDS.
//                           The leaf's idle wave sends every finished tile column of L_kk to memory and raises a column flag;
//   block (r, k), r > k     A[r,k] - sum_{j<k} L[r,j] L[k,j]^T: products with K = 128 per published block column (as many columns
//                           as are already there in one K loop), the sum in registers; then through LDS into the solve's layout
//                           and X = (that) inv(L_kk)^T by substitution over the tile columns BEHIND leaf k's column flags
//                           (follow): the solve ends a few microseconds after the leaf, not a substitution (16 us) after it.
//                           X goes to memory tile column by tile column with a flag per column (for the diagonal block of its
//                           row) and row_done[r] = k + 1 at the end (for the products to its right and below);
//   beside a trailing update (the look-ahead schedule, option lookahead_min) only the square has a workgroup per block; a block
//   row below it is one workgroup, left-looking: column k of row r is A[r,k] - L[r,0:k] L[k,0:k]^T in ONE product with K = 128 k
//   (it waits for row_done[k] = k only), then the solve against L_kk (trsm_sub).  Every resident workgroup holds a slot of the
//   update there, and a workgroup per block mostly waits.
//
// Tickets are dealt in START order (one atomic add per workgroup), and a ticket maps to a task by a fixed order in which a workgroup
// only ever waits for LOWER tickets -- which have started -- or, for the few tasks dealt ahead, for tickets at most a column's worth
// higher, which start as slots free: nothing depends on dispatch order or on all workgroups being resident together (a 4096-wide
// panel over 50 000 rows has twelve thousand).  The order: per block column the CRITICAL tasks -- the diagonal block and the two blocks
// under it, what the chain of leaves runs through -- and the BULK tasks (every other block of the column); the critical tasks of
// column k + ahead come right before the bulk tasks of column k.  ahead = 0 is plain column order ((0,0), (1,0), .. (1,1), (2,1), ..:
// the default).  ahead = 3 (option "chain_ahead", measurement only): the block the next leaf waits for has done all but its last
// products by the time the leaf before it ends, instead of starting them behind the hundreds of blocks of the column before -- the
// leaves of a tall panel then end 10 % earlier and the launch takes as long as before: it is bound by the products' throughput and by
// the drain behind the last column's workgroups, not by the chain of leaves (profiles/r06_chain_occupancy_n30000.txt).  Every wait is bounded (about 3 s): a workgroup that gives
// up raises the abort word, everybody leaves, and the host reports an error instead of hanging the GPU.
//
// Who gets the compute unit: two workgroups fit one.  The leaf, the block the next leaf waits for, and the block under it raise the
// unit's yield counter while they are on the critical path; the early products of the other blocks read that counter once per
// K step and sleep while it is up (as the trailing update does, gemm.hip).  A short panel alone on the chip runs ONE workgroup
// per compute unit (16 KB of unused LDS take the second one's room): a leaf then never shares its SIMDs.
//
// Hand-off form (MI355X_MICROARCH.md, inter-workgroup visibility): payload stored with sc1 (write-through) stores, every
// storing wave waits vmcnt(0), workgroup barrier (or, for the leaf's column flags, ONE wave stores and waits for itself), ONE
// lane stores the flag with sc1; the consumer's lane 0 polls the flag with sc1 loads, workgroup barrier, then every load of
// handed-off bytes is an sc1 load (buffer loads / LDS-DMA with aux = sc1, which go past the compute unit's L1).  No fences: a
// release would write back the dirty lines a trailing update keeps producing in the same L2.  Flags are 64-bit tags (launch
// base + progress), never reset; a flag written by several workgroups in turn (the column flags) gets its last value of one
// writer before the flag goes up that lets the next writer start.  Option "chain_verify": every hand-off carries a checksum.
//
// Resource shape: 512 threads at <= 128 registers and 73 KB of LDS -- what ONE retiring trailing-update workgroup frees, as
// the leaf kernel -- so a workgroup starts beside a full update.
#include "leaf_body.h"
#include <type_traits>

namespace {

struct ChainArgs {
    double *A; long lda;              // element (J0, J0): origin of the panel
    int n;                            // 128-column blocks of the panel
    int rows;                         // 128-row blocks from J0 to the end of the padded matrix (>= n)
    int sleep_rows;                   // panels of more block rows than this: no product yields its compute unit (option chain_sleep_rows)
    int n2;                           // n <= n2 <= rows: the first n2 block rows have one workgroup per BLOCK (the square always), the rest one per block row
    int nvalid;                       // rows of the matrix proper from J0 on (the rest is identity padding)
    double *linv;                     // inverses of the panel's diagonal blocks (n x 128 x 128)
    double *logdet;                   // 1 / L_ii, 128 per block
    int *info; int info_base;
    unsigned long long *flags;        // 16 words (one 128-byte line) apart: ticket, leaf_done, abort, diag_ready[32], row_done[32]
    unsigned long long tick0, tag0;
    int ahead;                        // block columns the critical tasks are dealt in front of the bulk tasks (chain_ticket_role)
    unsigned long long cols_tag;      // != 0: the block rows below the square wait for flags[F_COLS] to reach it (their part of the trailing update runs beside this launch)
    int *yield;
    int yield_below;                  // the block rows below the square raise their compute unit's yield counter too
    int leaf_preloaded;
    int leaf_factor, leaf_tiles;      // 1, 1 (run-time values: as constants they change the leaf's code, and its register allocation, for the worse)
    unsigned long *leaf_stamps;       // diagnostics (option "leaf_stamps"): phase times of the runner's leaves
    unsigned long long *stamps; int seq;      // diagnostics (option "chain_stamps"): {launch, code, task id << 24 | row << 8 | step, 100 MHz time} per event
    unsigned long long *vhash;        // option "chain_verify" (chain_kernel<true>): payload sums of the hand-offs, see VH_* below
};

// "chain_verify": every hand-off of the launch carries a checksum of its payload -- the sum of the 64-bit patterns of every double
// handed over, taken by the producer from the registers / LDS it stores from, published (sc1) before the flag; every consumer sums
// the bytes as they ARRIVED (the LDS images its LDS-DMA loads filled, the registers its buffer loads returned) and compares.  A
// stale or torn line changes a sum: VH_BAD counts mismatches, VH_CHECKS comparisons.  The words of a launch: row_done payloads
// [row][step] (1024 x 32), leaf payloads [block] (32), the two counters; zeroed by the host per launch.
constexpr int VH_ROWS = 1024;           // block rows that may have a workgroup per block
constexpr int VH_ROW = 0, VH_LEAF = VH_ROWS * 32, VH_BAD = VH_LEAF + 32, VH_CHECKS = VH_BAD + 1, VH_WORDS = VH_CHECKS + 1;

// ticket -> task of the resident panel kernel (also replayed on the host: fvgp_hip_debug_chain_ticket).  The first
// nsq = n n2 - n (n - 1) / 2 tickets are the blocks (row, k), k <= row < n2, of the first n2 block rows: kind 0 = diagonal block
// (row == k), 1 = block below it.  Per block column k the CRITICAL tasks are the diagonal block and rows k + 1 .. k + CHAIN_CRIT, the
// BULK tasks rows k + CHAIN_CRIT + 1 .. n2 - 1.  Order: critical columns 0 .. ahead, then for k = 0, 1, ..: bulk column k, critical
// column k + ahead + 1.  Later tickets: kind 2, a whole block row (row = n2 + ticket - nsq, then every `stride`-th row).
// What a task waits for is always in a column before its own, or the leaf of its own column:
//   (row, k), row > k : (row, j) and (k, j) for j < k, and the leaf of (k, k);   (k, k): (k, j), j < k;
//   a block row: (k, j) for j < k < n and every leaf.
// So a BULK task of column k only waits for lower tickets (every critical task of columns <= k + ahead and every bulk task of the
// columns before k come first); with ahead = 0 so does a critical task.  A critical task dealt ahead may wait for bulk tickets of
// the `ahead` columns in between: they start as slots free, and at most (ahead + 1) (CHAIN_CRIT + 1) critical tasks are ahead at any
// time (tests/test_host_logic.py replays the order with as few as 16 slots).
constexpr int CHAIN_CRIT = 2;
__host__ __device__ inline void chain_ticket_role(const int n, const int n2, const int ahead, const int t, int *kind, int *row, int *col) {
    int pos = t;
    int kc = 0;                                   // next critical column to place
    for (int k = -1; k < n; ++k) {
        if (k >= 0) {                             // bulk column k
            const int below = n2 - 1 - k, nc = below < CHAIN_CRIT ? below : CHAIN_CRIT, cnt = below - nc;
            if (pos < cnt) { *col = k; *row = k + nc + 1 + pos; *kind = 1; return; }
            pos -= cnt;
        }
        // critical columns up to k + ahead + 1 (k = -1: columns 0 .. ahead)
        for (; kc < n && kc <= k + ahead + 1; ++kc) {
            const int below = n2 - 1 - kc, cnt = 1 + (below < CHAIN_CRIT ? below : CHAIN_CRIT);
            if (pos < cnt) { *col = kc; *row = kc + pos; *kind = pos == 0 ? 0 : 1; return; }
            pos -= cnt;
        }
    }
    *kind = 2; *row = n2 + pos; *col = -1;
}

constexpr int FL = 16;                // 64-bit words between two flags
constexpr int F_TICKET = 0, F_LEAF = 1, F_ABORT = 2, F_COLS = 40, F_LCOL = 41, F_XCOL = 42, F_DUMMY = 75, F_ROW = 96;      // F_XCOL: one per block row of the square (32), F_ROW: one per block row with a workgroup per block (VH_ROWS)
constexpr int FLAG_LINES = F_ROW + VH_ROWS;
constexpr int IMGD = 128 * 16;        // doubles of one operand image (128 rows x 16 k)

__device__ __forceinline__ unsigned long long flag_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void chain_stamp(const ChainArgs &g, const int code, const int t, const int row, const int step) {
    if (g.stamps && threadIdx.x == 0) {
        const unsigned long long i = atomicAdd(g.stamps, 1ull);
        if (i < (1ull << 20)) {
            unsigned long long *e = g.stamps + 8 + 4 * i;
            e[0] = (unsigned long long)g.seq; e[1] = (unsigned long long)code; e[2] = ((unsigned long long)t << 24) | ((unsigned long long)row << 8) | (unsigned long long)step;
            e[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// lane 0 polls until *flag has reached `need`; false when the launch is being abandoned
// (`relaxed`: the waiter is not what the next leaf waits for -- it polls four times less often: hundreds of workgroups of a wide panel wait at a time)
__device__ __forceinline__ bool chain_wait(const ChainArgs &g, const int flag, const unsigned long long need, int *s_ok, const bool relaxed = false) {
    if (threadIdx.x == 0) {
        int ok = 1, it = 0;
        unsigned long long t0 = 0;
        const unsigned long long *p = g.flags + (long)flag * FL;
        while ((long long)(flag_load(p) - need) < 0) {
            if (relaxed) __builtin_amdgcn_s_sleep(40); else __builtin_amdgcn_s_sleep(8);
            if ((++it & 127) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();      // 100 MHz
                if (t0 == 0) t0 = now;
                if (now - t0 > 300000000ull || flag_load(g.flags + F_ABORT * FL) == g.tag0) {
                    flag_store(g.flags + F_ABORT * FL, g.tag0);
                    atomicCAS(g.info, 0, 0x7fffffff);
                    ok = 0;
                    break;
                }
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    const int ok = *s_ok;
    __syncthreads();
    return ok != 0;
}

// lane 0 polls TWO progress flags at once until both have reached `need`, and hands back how far the SLOWER one has come (its value
// minus `base`, at most `cap`): the row flags of a block's two operand rows.  One round trip per poll for both flags and one
// barrier pair, where two chain_waits and a separate reading of both flags cost six barriers and four dependent round trips (5-6 us
// in front of every product of a block that follows its columns' publication one by one).  -1: the launch is being abandoned.
__device__ __forceinline__ int chain_wait2(const ChainArgs &g, const int flag_a, const int flag_b, const unsigned long long base, const unsigned long long need,
                                           const int cap, int *s_ok, const bool relaxed) {
    if (threadIdx.x == 0) {
        int it = 0;
        unsigned long long t0 = 0;
        const unsigned long long *pa = g.flags + (long)flag_a * FL, *pb = g.flags + (long)flag_b * FL;
        long long m;
        for (;;) {
            const unsigned long long a = flag_load(pa), b = flag_load(pb);
            const long long da = (long long)(a - base), db = (long long)(b - base);
            m = da < db ? da : db;
            if (m >= (long long)(need - base)) break;
            if (relaxed) __builtin_amdgcn_s_sleep(40); else __builtin_amdgcn_s_sleep(8);
            if ((++it & 127) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();      // 100 MHz
                if (t0 == 0) t0 = now;
                if (now - t0 > 300000000ull || flag_load(g.flags + F_ABORT * FL) == g.tag0) {
                    flag_store(g.flags + F_ABORT * FL, g.tag0);
                    atomicCAS(g.info, 0, 0x7fffffff);
                    m = -1;
                    break;
                }
            }
        }
        *s_ok = m < 0 ? -1 : (m > cap ? cap : (int)m);
    }
    __syncthreads();
    const int r = *s_ok;
    __syncthreads();
    return r;
}

// every store of this workgroup has left, then one lane raises the flag
__device__ __forceinline__ void chain_publish(const ChainArgs &g, const int flag, const unsigned long long value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) flag_store(g.flags + (long)flag * FL, value);
}

__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); }

__device__ __forceinline__ unsigned long long bits_of(const double v) { return (unsigned long long)__double_as_longlong(v); }
// sum over the workgroup (chain_verify only)
__device__ __forceinline__ unsigned long long wg_sum(const unsigned long long v, unsigned long long *s_acc) {
    __syncthreads();
    if (threadIdx.x == 0) *s_acc = 0ull;
    __syncthreads();
    atomicAdd(s_acc, v);
    __syncthreads();
    const unsigned long long r = *s_acc;
    __syncthreads();
    return r;
}
__device__ __forceinline__ void verify_compare(const ChainArgs &g, const unsigned long long got, const unsigned long long want) {
    if (threadIdx.x == 0) {
        atomicAdd(g.vhash + VH_CHECKS, 1ull);
        if (got != want) atomicAdd(g.vhash + VH_BAD, 1ull);
    }
}

// acc (128 x 128 over eight waves, 2 x 4, 64 x 32 each) = sum_k Aop[row][k] Bop[col][k], k < 16 nk.  Both operands k-minor,
// row-major; the K loop of the trailing update (gemm.hip): unpadded [128][16] images with XOR-swizzled 16-byte chunks filled
// by LDS-DMA, lane group q owns k = 4q .. 4q+3 of a step, every address loop-invariant.  All loads sc1.
// With C given the sum starts at -C (entries above the diagonal of a `lower` block at 0, never read): the caller stores -acc = C - sum.
// VERIFY: *bsum (*asum, if given) receives the sum of the bit patterns of every double of the B (A) operand as it landed in LDS (per thread; the caller adds them up)
// KEEP: acc goes on from where the last product left it (nothing zeroed, C not read)
template <bool VERIFY = false, bool KEEP = false>
__device__ __forceinline__ void product(double4_t (&acc)[4][2], const double *Aop, const long lda, const double *Bop, const long ldb,
                                        const int nk, double *smem, const double *C = nullptr, const long ldc = 0, const bool lower = false,
                                        unsigned long long *bsum = nullptr, unsigned long long *asum = nullptr, const int *ybase = nullptr) {
    typedef __attribute__((address_space(3))) void lds_void;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                  // opaque: the addressing of one product is not kept alive across the others
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Aop)), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Bop)), 0, 0xffffffff, 0x00020000);
    int voa[2], vob[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = p * 64 + (tid >> 3), kc = ((tid & 7) ^ swz(row)) * 2;
        voa[p] = (int)(((long)row * lda + kc) * 8);
        vob[p] = (int)(((long)row * ldb + kc) * 8);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // fragment addresses: ONE register per operand and half-step, everything else (row tile, LDS buffer, operand image) is an
    // immediate offset of the read
    const int c0 = (2 * q) ^ swz(r);
    const int fa0 = (wm * 64 + r) * 16 + 2 * c0, fa1 = (wm * 64 + r) * 16 + 2 * (c0 ^ 1);
    const int fb0 = IMGD + (wn * 32 + r) * 16 + 2 * c0, fb1 = IMGD + (wn * 32 + r) * 16 + 2 * (c0 ^ 1);
    if constexpr (!KEEP) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    if (!KEEP && C) {
        const double *cb = C + (long)(wm * 64 + q) * ldc + wn * 32 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = wm * 64 + i * 16 + q + 4 * v, col = wn * 32 + j * 16 + r;
                    // (all 32 loads in flight together: a predicated load would wait for the one before it; what lies above
                    // the diagonal is read and dropped)
                    const double cv = ld_sc1(cb + (long)(i * 16 + 4 * v) * ldc + j * 16);
                    acc[i][j][v] = (!lower || col <= row) ? -cv : 0.0;
                }
    }
    auto dma = [&](auto bufc, const int soff) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            lds_void *da = (lds_void *)&smem[(BUF * 2 + 0) * IMGD + (p * 64 + wave_u * 8) * 16];
            lds_void *db = (lds_void *)&smem[(BUF * 2 + 1) * IMGD + (p * 64 + wave_u * 8) * 16];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_src, da, 16, voa[p], soff, 0, 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_src, db, 16, vob[p], soff, 0, 16);
        }
    };
    int soff = 0;
    // ybase: this product is not what the panel's next leaf waits for -- like the trailing update (gemm.hip, YIELD) every wave reads its
    // compute unit's yield counter once per K step with a scalar load and sleeps while a leaf, or the solve the next leaf waits for, runs
    // on this compute unit (two workgroups of this kernel share one; beside sixteen MFMA waves a leaf takes 45 us instead of 24)
    const int *yp = ybase ? cu_yield_slot(const_cast<int *>(ybase)) : nullptr;
    int ybudget = 64;
    auto kstep = [&](auto curc, const bool more) {
        constexpr int CUR = decltype(curc)::value;
        if (more) { soff += 128; dma(std::integral_constant<int, CUR ^ 1>{}, soff); }
        const double *ps = &smem[CUR * 2 * IMGD];
        int yv = 0;
        if (yp) asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(yv) : "s"(yp));
        if constexpr (VERIFY) {           // this step's B image, 2048 doubles: four per thread (a sum does not mind the swizzle)
            const unsigned long long *pb = reinterpret_cast<const unsigned long long *>(ps + IMGD) + 4 * tid;
            *bsum += (pb[0] + pb[1]) + (pb[2] + pb[3]);
            if (asum) { const unsigned long long *pa = reinterpret_cast<const unsigned long long *>(ps) + 4 * tid; *asum += (pa[0] + pa[1]) + (pa[2] + pa[3]); }
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            double2_t a2[4], b2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a2[i] = *reinterpret_cast<const double2_t *>(ps + (hf ? fa1 : fa0) + i * 256);
#pragma unroll
            for (int j = 0; j < 2; ++j) b2[j] = *reinterpret_cast<const double2_t *>(ps + (hf ? fb1 : fb0) + j * 256);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i][s], b2[j][s], acc[i][j], 0, 0, 0);
        }
        if (yp) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(yv) :: "memory");
            while (yv != 0 && ybudget > 0) {
                --ybudget;
                __builtin_amdgcn_s_sleep(127);
                asm volatile("s_load_dword %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(yv) : "s"(yp) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next step's image has landed
        __syncthreads();
    };
    dma(std::integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(std::integral_constant<int, 0>{}, true);
        kstep(std::integral_constant<int, 1>{}, kt + 2 < nk);
    }
    if (kt < nk) kstep(std::integral_constant<int, 0>{}, false);
}

template <bool SC1>
__device__ __forceinline__ void st_out(double *p, double v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// C = acc or (NEG) -acc; `lower`: the block is a diagonal block, only entries on / below its diagonal are written (the strict upper
// triangle of the matrix is nobody's)
template <bool NEG, bool SC1>
__device__ __forceinline__ void epilogue(const double4_t (&acc)[4][2], double *C, const long ldc, const bool lower = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    double *cb = C + (long)(wm * 64 + q) * ldc + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = wm * 64 + i * 16 + q + 4 * v, col = wn * 32 + j * 16 + r;
                if (lower && col > row) continue;
                st_out<SC1>(cb + (long)(i * 16 + 4 * v) * ldc + j * 16, NEG ? -acc[i][j][v] : acc[i][j][v]);
            }
}

// acc <- -C (the start of `product` with C given, on its own: a block's workgroup loads C before it waits for its operands)
__device__ __forceinline__ void load_c_neg(double4_t (&acc)[4][2], const double *C, const long ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const double *cb = C + (long)(wm * 64 + q) * ldc + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[i][j][v] = -ld_sc1(cb + (long)(i * 16 + 4 * v) * ldc + j * 16);
}

// The block C - sum as `product` leaves it (acc = -(C - sum), eight waves of 64 x 32) -> the solve's layout (xt: wave w owns rows
// 16 w .., lane (r, q) columns 16 t + 4 q .. + 3 of row r) through LDS, 64 rows at a time (a row of the staging buffer is 144 doubles:
// the four lane groups of a wave, four rows apart, land on alternating halves of the banks) -- instead of a write-through store of the
// block, a drain and sixteen loads back (4-5 us on the path of the block the next leaf waits for).  The staging buffer takes all 72 KB.
constexpr int XSTG = 144;
__device__ __forceinline__ void acc_to_xt(const double4_t (&acc)[4][2], double4_t (&xt)[8], double *smem) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave_u & 3;
    auto put = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) smem[(i * 16 + q + 4 * v) * XSTG + wn * 32 + j * 16 + r] = -acc[i][j][v];
    };
    auto get = [&]() {
        const double *row = &smem[((wave_u & 3) * 16 + r) * XSTG + 4 * q];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double2_t a = *reinterpret_cast<const double2_t *>(row + 16 * t), b = *reinterpret_cast<const double2_t *>(row + 16 * t + 2);
            xt[t] = (double4_t){a[0], a[1], b[0], b[1]};
        }
    };
    // (the same four barriers on both paths: rows 0 .. 63 go through the buffer first, then rows 64 .. 127)
    if (wave_u < 4) { put(); __syncthreads(); get(); __syncthreads(); __syncthreads(); __syncthreads(); }
    else { __syncthreads(); __syncthreads(); put(); __syncthreads(); get(); __syncthreads(); }
}

// X = A inv(L)^T in place for one 128 x 128 block A, by substitution over the eight 16-column tiles of the lower block L with the
// inverses of L's diagonal tiles only (what the leaf leaves: `dinv`, 8 x 256 doubles, zeros above the diagonals) -- no
// 128 x 128 inverse anywhere on the chain.  L's strictly lower tiles and the tile inverses sit in LDS as the leaf's 36 packed
// tiles; a wave owns 16 rows of A and needs nobody else: X_t^T = inv(L_tt) (A_t^T - sum_{s<t} L_ts X_s^T), every X_s^T kept in
// registers in the MFMA accumulator layout, which IS the next product's B operand.  The sixteen columns of a tile are dealt to
// the accumulator rows by c = 4 (i mod 4) + i div 4, so that a lane holds four CONSECUTIVE columns of its row: A is read and X
// written in 32-byte pieces, L's fragments are 16-byte LDS reads.
// this wave's sixteen rows of a 128 x 128 block: lane (r, q) takes columns 16 t + 4 q .. + 3 of row r, t = 0 .. 7 (sc1 loads, all
// sixteen in flight; issued BEFORE the wait for the leaf the solve depends on -- the block itself was final a step earlier)
__device__ __forceinline__ void trsm_load(double4_t (&xt)[8], const double *Ablk, const long lda) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Ablk + (long)wave_u * 16 * lda)), 0, 0xffffffff, 0x00020000);
    const int vo = (int)(((long)r * lda + 4 * q) * 8);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(a_src, vo, t * 128, 16);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(a_src, vo, t * 128 + 16, 16);
        double2_t d0, d1;
        __builtin_memcpy(&d0, &lo, 16); __builtin_memcpy(&d1, &hi, 16);
        xt[t] = (double4_t){d0[0], d0[1], d1[0], d1[1]};
    }
}

// VERIFY: *lsum receives this thread's share of the sum of the bit patterns of L's strictly lower tiles and the tile inverses as they
// arrived in LDS (the leaf hand-off's payload)
template <bool SC1_STORE, bool VERIFY = false>
__device__ __forceinline__ void trsm_sub(double4_t (&xt)[8], double *Ablk, const long lda, const double *L, const long ldl, const double *dinv, double *sT,
                                         unsigned long long *lsum = nullptr) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Ablk + (long)wave_u * 16 * lda)), 0, 0xffffffff, 0x00020000);
    const int vo = (int)(((long)r * lda + 4 * q) * 8);
    {   // L's 28 strictly lower tiles -> packed tiles by LDS-DMA: a wave instruction lands 1 KB = rows 8 u .. 8 u + 7 of one tile
        // (lane l at 16 l bytes: row 8 u + (l >> 3), position l & 7 of the row's eight 16-byte pieces, which holds piece
        // (l & 7) ^ ((row >> 1) & 7) of the tile's row: the tiles' own column swizzle); 56 instructions, 7 per wave, all in flight
        // (as register loads two at a time the block took seven memory round trips)
        typedef __attribute__((address_space(3))) void lds_void;
        const __amdgpu_buffer_rsrc_t l_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(L)), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int hx = wave_u + 8 * it;                      // half-tile 0 .. 55: strictly lower tile hx >> 1, rows 8 (hx & 1) ..
            const int p = hx >> 1, u = hx & 1;
            int ti = 1;
            while (ti * (ti + 1) / 2 <= p) ++ti;                 // strictly lower tiles in row-major order: p = ti (ti - 1) / 2 + tj
            const int tj = p - ti * (ti - 1) / 2;
            const int a = 8 * u + (lane >> 3);
            const int piece = (lane & 7) ^ ((a >> 1) & 7);
            const int voff = (int)(((long)(16 * ti + a) * ldl + 16 * tj + 2 * piece) * 8);
            lds_void *dst = (lds_void *)&sT[tix(ti, tj) + 8 * u * 16];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(l_src, dst, 16, voff, 0, 0, 16);
        }
    }
    {   // the diagonal tiles <- their inverses
        const __amdgpu_buffer_rsrc_t d_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(dinv)), 0, 0xffffffff, 0x00020000);
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(d_src, tid * 32, 0, 16);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(d_src, tid * 32 + 16, 0, 16);
        const int t = tid >> 6, a = (tid >> 2) & 15, b = (tid & 3) * 4;
        double *T = &sT[tix(t, t)];
        *reinterpret_cast<u32x4 *>(&T[el(a, b)]) = lo;
        *reinterpret_cast<u32x4 *>(&T[el(a, b + 2)]) = hi;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA'd tiles have landed
    __syncthreads();
    if constexpr (VERIFY) {                              // 36 x 256 doubles: eighteen per thread
        const unsigned long long *pt = reinterpret_cast<const unsigned long long *>(sT);
        unsigned long long a = 0ull;
#pragma unroll
        for (int i = 0; i < 18; ++i) a += pt[tid + 512 * i];
        *lsum = a;
    }
    const int pr = 4 * (r & 3) + (r >> 2);           // the row of a 16 x 16 tile this lane supplies as MFMA row r
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        double4_t a0 = xt[t], a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0}, a3 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < t; ++s) {
            const double *T = &sT[tix(t, s)];
            const double2_t l01 = *reinterpret_cast<const double2_t *>(&T[el(pr, 4 * q)]);
            const double2_t l23 = *reinterpret_cast<const double2_t *>(&T[el(pr, 4 * q + 2)]);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[0], xt[s][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[1], xt[s][1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[0], xt[s][2], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[1], xt[s][3], a3, 0, 0, 0);
        }
        const double4_t rr = (a0 + a1) + (a2 + a3);
        const double *D = &sT[tix(t, t)];
        const double2_t d01 = *reinterpret_cast<const double2_t *>(&D[el(pr, 4 * q)]);
        const double2_t d23 = *reinterpret_cast<const double2_t *>(&D[el(pr, 4 * q + 2)]);
        double4_t x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d01[0], rr[0], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d01[1], rr[1], x1, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d23[0], rr[2], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d23[1], rr[3], x1, 0, 0, 0);
        xt[t] = x0 + x1;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        u32x4 lo, hi;
        const double2_t d0 = {xt[t][0], xt[t][1]}, d1 = {xt[t][2], xt[t][3]};
        __builtin_memcpy(&lo, &d0, 16); __builtin_memcpy(&hi, &d1, 16);
        // (the tile's byte offset goes into the instruction's immediate, NOT the scalar offset operand: with an SGPR there the compiler
        // assumes that a 16-byte store's data registers may be overwritten by the very next vector instruction -- the documented
        // exception of the ">64-bit store data" hazard -- and on gfx950 they may not: the checksummed build of this kernel put the
        // sum for the next store right behind one, and the last lanes of that store went out with the NEW register contents, in one
        // run of ten (tools/chain_verify_stress.py).  With no SGPR offset the hazard recogniser inserts the wait state.)
        __builtin_amdgcn_raw_buffer_store_b128(lo, a_src, vo + t * 128, 0, SC1_STORE ? 16 : 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, a_src, vo + t * 128 + 16, 0, SC1_STORE ? 16 : 0);
    }
}

// the 36 lower 16 x 16 tiles of a 128 x 128 diagonal block, dealt round-robin to the eight waves, NEGATED, in the MFMA accumulator
// layout (syrk_follow adds X X^T to them): 20 eight-byte sc1 loads per lane, all in flight at once
__device__ __forceinline__ void diag_load(double4_t (&res)[5], const double *D, const long ldd) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t d_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(D)), 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int p = wave_u + 8 * i;
        if (p >= NT) p = NT - 1;                                 // (waves 4 .. 7 have four tiles: the fifth load is dropped)
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= p) ++a;
        const int b = p - a * (a + 1) / 2;
#pragma unroll
        for (int v = 0; v < 4; ++v) {                            // (above the diagonal: read and dropped)
            const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(d_src, (int)(((long)(q + 4 * v) * ldd + r) * 8), (int)(((long)(16 * a) * ldd + 16 * b) * 8), 16);
            double d;
            __builtin_memcpy(&d, &raw, 8);
            res[i][v] = -d;
        }
    }
}

// A block row of the panel's SQUARE solves its block against L_jj BEHIND the leaf that is still factoring it: the leaf's idle wave
// sends every finished 16-column tile column of L_jj (and the inverse of its diagonal tile) to memory and raises F_LCOL (leaf_body.h,
// publish_column); tile column t of the solve, X_t^T = inv(L_tt) (A_t^T - sum_{s<t} L_ts X_s^T) -- trsm_sub's step t -- needs row t of
// L's tiles and inv(L_tt) only, which are there once column t is flagged.  So the solve runs two tile columns behind the
// factorisation and is done a few microseconds after the leaf instead of a whole substitution (16 us) after it.  The solved tile
// columns are flagged one by one in turn (xcol: column t - 1 once the wait for column t's loads has also covered its stores) for
// the workgroup that keeps this row's diagonal block up to date (syrk_follow).
// LDS: two buffers of eight tiles for row t of L (+ the tile inverse).  False: the launch is being abandoned.
// The arithmetic, operation by operation, is trsm_sub's: same bits.
template <bool VERIFY>
__device__ __forceinline__ bool follow(const ChainArgs &g, double4_t (&xt)[8], double *Ablk, const long lda, const double *L, const long ldl,
                                       const double *dinv, const unsigned long long col_need0, unsigned long long *xcol, double *smem, int *s_ok,
                                       unsigned long long *lsum) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void lds_void;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(Ablk + (long)wave_u * 16 * lda)), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t l_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(L)), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(dinv)), 0, 0xffffffff, 0x00020000);
    const int vo = (int)(((long)r * lda + 4 * q) * 8);
    const int pr = 4 * (r & 3) + (r >> 2);           // the row of a 16 x 16 tile this lane supplies as MFMA row r
    unsigned long long seen = 0ull;                  // thread 0: the column flag as last read
    [[maybe_unused]] unsigned long long vs = 0ull;
    // row t of L's tiles, (t, 0) .. (t, t-1), and inv(L_tt) -> buffer t & 1, tiles 0 .. t: 2 t + 2 half-tiles of 1 KB, one LDS-DMA instruction each
    auto fetch = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        double *buf = smem + (t & 1) * 8 * TSZ;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hx = wave_u + 8 * it;
            if (hx < 2 * t + 2) {
                const int sx = hx >> 1, u = hx & 1;
                const int a = 8 * u + (lane >> 3);
                const int piece = (lane & 7) ^ ((a >> 1) & 7);
                lds_void *dst = (lds_void *)&buf[sx * TSZ + 8 * u * 16];
                if (sx < t) __builtin_amdgcn_raw_ptr_buffer_load_lds(l_src, dst, 16, (int)(((long)(16 * t + a) * ldl + 16 * sx + 2 * piece) * 8), 0, 0, 16);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(d_src, dst, 16, (t * 256 + a * 16 + 2 * piece) * 8, 0, 0, 16);
            }
        }
    };
    // navail: tile columns of L known to be flagged (thread 0's last poll); ahead: this column's tiles were requested a column ago (the
    // solver is catching up with a leaf that is ahead of it: the round trip of the loads hides under the column before)
    int navail = 0;
    bool ahead = false;
    auto column = [&](auto tc) -> bool {
        constexpr int t = decltype(tc)::value;
        double *buf = smem + (t & 1) * 8 * TSZ;
        if (!ahead) {
            if (tid == 0) {
                int ok = 1;
                const unsigned long long need = col_need0 + t + 1;
                if ((long long)(seen - need) < 0) {
                    const unsigned long long *p = g.flags + (long)F_LCOL * FL;
                    int it = 0;
                    unsigned long long t0 = 0;
                    while ((long long)((seen = flag_load(p)) - need) < 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((++it & 255) == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();      // 100 MHz
                            if (t0 == 0) t0 = now;
                            if (now - t0 > 300000000ull || flag_load(g.flags + F_ABORT * FL) == g.tag0) {
                                flag_store(g.flags + F_ABORT * FL, g.tag0);
                                atomicCAS(g.info, 0, 0x7fffffff);
                                ok = 0;
                                break;
                            }
                        }
                    }
                }
                const unsigned long long have = seen - col_need0;
                s_ok[0] = ok ? (have > 8ull ? 8 : (int)have) : -1;
            }
            __syncthreads();                     // (also: nobody reads buffer t & 1 of two columns ago any more)
            navail = s_ok[0];
            if (navail < 0) return false;
            fetch(tc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this column's tiles have landed -- and the last column's solved tile has left
            __syncthreads();
            if (t > 0 && tid == 0) flag_store(xcol, col_need0 + t);
        } else {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the tiles asked for a column ago have landed (behind them: the two stores of the last solved tile)
            __syncthreads();
            if (t > 1 && tid == 0) flag_store(xcol, col_need0 + t - 1);
        }
        ahead = false;
        if constexpr (t < 7) {
            if (navail >= t + 2) { fetch(std::integral_constant<int, t + 1>{}); ahead = true; }
        }
        if constexpr (VERIFY) {
            const unsigned long long *pt = reinterpret_cast<const unsigned long long *>(buf);
            for (int e = tid; e < (t + 1) * TSZ; e += 512) vs += pt[e];
        }
        {
            double4_t a0 = xt[t], a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0}, a3 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < t; ++s) {
                const double *T = &buf[s * TSZ];
                const double2_t l01 = *reinterpret_cast<const double2_t *>(&T[el(pr, 4 * q)]);
                const double2_t l23 = *reinterpret_cast<const double2_t *>(&T[el(pr, 4 * q + 2)]);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[0], xt[s][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l01[1], xt[s][1], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[0], xt[s][2], a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l23[1], xt[s][3], a3, 0, 0, 0);
            }
            const double4_t rr = (a0 + a1) + (a2 + a3);
            const double *D = &buf[t * TSZ];
            const double2_t d01 = *reinterpret_cast<const double2_t *>(&D[el(pr, 4 * q)]);
            const double2_t d23 = *reinterpret_cast<const double2_t *>(&D[el(pr, 4 * q + 2)]);
            double4_t x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d01[0], rr[0], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d01[1], rr[1], x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d23[0], rr[2], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d23[1], rr[3], x1, 0, 0, 0);
            xt[t] = x0 + x1;
        }
        {   // (no scalar offset on a 16-byte store: see trsm_sub)
            u32x4 lo, hi;
            const double2_t d0 = {xt[t][0], xt[t][1]}, d1 = {xt[t][2], xt[t][3]};
            __builtin_memcpy(&lo, &d0, 16); __builtin_memcpy(&hi, &d1, 16);
            __builtin_amdgcn_raw_buffer_store_b128(lo, a_src, vo + t * 128, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(hi, a_src, vo + t * 128 + 16, 0, 16);
        }
        return true;
    };
    if (!column(std::integral_constant<int, 0>{})) return false;
    if (!column(std::integral_constant<int, 1>{})) return false;
    if (!column(std::integral_constant<int, 2>{})) return false;
    if (!column(std::integral_constant<int, 3>{})) return false;
    if (!column(std::integral_constant<int, 4>{})) return false;
    if (!column(std::integral_constant<int, 5>{})) return false;
    if (!column(std::integral_constant<int, 6>{})) return false;
    if (!column(std::integral_constant<int, 7>{})) return false;
    if constexpr (VERIFY) *lsum = vs;
    return true;
}

// The workgroup that will factor diagonal block `row` keeps that block in registers for the whole launch (`res`: its 36 lower
// 16 x 16 tiles dealt round-robin to the eight waves, negated, as diag_load leaves them) and subtracts X X^T for every solved block
// X = L[row, j] of its block row as the solver produces it, one tile column (K = 16) at a time behind the solver's column flags:
// the 128 x 16 tile column comes in by LDS-DMA as eight packed tiles (two buffers), each wave reads both operands of its tiles from
// there -- the accumulator layout of X^T is both MFMA operands of X X^T.  False: the launch is being abandoned.
// VERIFY: *xsum receives this thread's share of the sum of the bit patterns of the block as it arrived.
template <bool VERIFY>
__device__ __forceinline__ bool syrk_follow(const ChainArgs &g, double4_t (&res)[5], const double *X, const long ldx, const unsigned long long col_need0,
                                            const unsigned long long *xcol, double *smem, int *s_ok, unsigned long long *xsum) {
    typedef __attribute__((address_space(3))) void lds_void;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(uniform_ptr(X)), 0, 0xffffffff, 0x00020000);
    int ta[5], tb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave_u + 8 * i;
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= p) ++a;
        ta[i] = a; tb[i] = p - a * (a + 1) / 2;
    }
    // this lane's share of a tile column: wave w lands tile w (rows 16 w ..), half u: row 8 u + (lane >> 3), its eight 16-byte pieces swizzled as in el()
    int voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int a = 8 * u + (lane >> 3);
        const int piece = (lane & 7) ^ ((a >> 1) & 7);
        voff[u] = (int)(((long)(16 * wave_u + a) * ldx + 2 * piece) * 8);
    }
    const int o0 = el(r, 4 * q), o1 = el(r, 4 * q + 2);
    unsigned long long seen = 0ull;
    [[maybe_unused]] unsigned long long vs = 0ull;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (tid == 0) {
            int ok = 1;
            const unsigned long long need = col_need0 + t + 1;
            if ((long long)(seen - need) < 0) {
                int it = 0;
                unsigned long long t0 = 0;
                while ((long long)((seen = flag_load(xcol)) - need) < 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++it & 255) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();      // 100 MHz
                        if (t0 == 0) t0 = now;
                        if (now - t0 > 300000000ull || flag_load(g.flags + F_ABORT * FL) == g.tag0) {
                            flag_store(g.flags + F_ABORT * FL, g.tag0);
                            atomicCAS(g.info, 0, 0x7fffffff);
                            ok = 0;
                            break;
                        }
                    }
                }
            }
            *s_ok = ok;
        }
        __syncthreads();                         // (also: nobody reads buffer t & 1 of two columns ago any more)
        if (*s_ok == 0) return false;
        double *buf = smem + (t & 1) * 8 * TSZ;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_src, (lds_void *)&buf[wave_u * TSZ + 8 * u * 16], 16, voff[u], t * 128, 0, 16);      // (the column's byte offset as the scalar offset: an instruction offset would move the LDS address too)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (VERIFY) {
            const unsigned long long *pt = reinterpret_cast<const unsigned long long *>(buf);
            vs += (pt[tid] + pt[tid + 512]) + (pt[tid + 1024] + pt[tid + 1536]);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (wave_u + 8 * i < NT) {
                const double *Ta = &buf[ta[i] * TSZ], *Tb = &buf[tb[i] * TSZ];
                const double2_t a01 = *reinterpret_cast<const double2_t *>(&Ta[o0]), a23 = *reinterpret_cast<const double2_t *>(&Ta[o1]);
                const double2_t b01 = *reinterpret_cast<const double2_t *>(&Tb[o0]), b23 = *reinterpret_cast<const double2_t *>(&Tb[o1]);
                res[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[0], b01[0], res[i], 0, 0, 0);
                res[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[1], b01[1], res[i], 0, 0, 0);
                res[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[0], b23[0], res[i], 0, 0, 0);
                res[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[1], b23[1], res[i], 0, 0, 0);
            }
        }
    }
    if constexpr (VERIFY) *xsum = vs;
    return true;
}

// `res` (syrk_follow) -> the leaf's packed LDS tiles: the last update of a diagonal block and its factorisation share a workgroup,
// the block never goes back to memory in between
__device__ __forceinline__ void diag_to_tiles(const double4_t (&res)[5], double *smem) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();                                              // everybody has read the tile columns: D's tiles may land
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave_u + 8 * i;
        if (p < NT) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= p) ++a;
            const int b = p - a * (a + 1) / 2;
            double *T = &smem[tix(a, b)];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ii = q + 4 * v;
                T[el(ii, r)] = (a == b && r > ii) ? 0.0 : -res[i][v];
            }
        }
    }
    __syncthreads();
}

template <bool VERIFY>
__global__ __launch_bounds__(512, 4) void chain_kernel(ChainArgs g) {
    __shared__ double smem[NT * TSZ + LEAF_SRD];      // the leaf's packed triangle + its small arrays (75,792 B); the products use the first 64 KB
    __shared__ int s_i[2];
    __shared__ unsigned long long s_acc;         // chain_verify: workgroup sums
    const int tid = threadIdx.x;
    if (tid == 0) s_i[0] = (int)(atomicAdd(g.flags + F_TICKET * FL, 1ull) - g.tick0);
    __syncthreads();
    const int t = s_i[0];
    const int n = g.n;
    const unsigned long long tag = g.tag0;
    chain_stamp(g, 0, t, 0, 0);
    double4_t acc[4][2];

    // tickets of the first n2 block rows, block column after block column: (0,0), (1,0), .. (n2-1,0), (1,1), (2,1), ..  A diagonal block
    // waits for the blocks left of it in its row, a block (row, k) for blocks of the columns before k and for leaf k: always a lower ticket
    const int n2 = g.n2;
    const int nsq = n * n2 - n * (n - 1) / 2;
    int bk = 0, brow = 0;
    if (t < nsq) { int kind; chain_ticket_role(n, n2, g.ahead, t, &kind, &brow, &bk); }
    if (t < nsq && brow != bk) {
        // ---- block (row, k) of the panel below the diagonal: A[row,k] - sum_{j<k} L[row,j] L[k,j]^T, one K = 128 product per
        //      block column j as soon as both operands are published (the sum stays in registers), then the solve against L_kk
        //      BEHIND leaf k (follow), then publication as row_done[row] = k + 1 ----
        const int row = brow, k = bk;
        double *Ar = g.A + (long)row * 128 * g.lda;
        const double *Ak = g.A + (long)k * 128 * g.lda;
        unsigned long long *xcol = g.flags + (long)(row < n ? F_XCOL + row : F_DUMMY) * FL;      // (only a row of the square has a diagonal block behind it)
        int *yslot = nullptr;
        // (the block the next leaf waits for goes first wherever it shares a SIMD: two workgroups of this kernel fit a compute unit, and in a
        // wide panel every compute unit has two)
        if (row == k + 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
        // (a row below the square: its part of the trailing update may still be running beside this launch)
        if (row >= n && g.cols_tag && !chain_wait(g, F_COLS, g.cols_tag, &s_i[1])) return;
        if (k > 0) {
            load_c_neg(acc, Ar + k * 128, g.lda);      // (before the first wait: the block itself has been final since the last trailing update)
            for (int j = 0; j < k;) {
                // every block column already published on both rows goes into ONE product (they are contiguous along K: a workgroup that
                // starts late -- most do, a tall panel has twenty times more of them than slots -- restarts its K loop once, not per column);
                // both rows' flags in one poll (chain_wait2)
                int m = chain_wait2(g, F_ROW + row, F_ROW + k, tag, tag + j + 1, k, &s_i[1], row != k + 1);
                if (m < 0) { if (yslot) atomicAdd(yslot, -1); return; }
                // the LAST product of a block is what its solve behind leaf k waits for (and with it, one way or another, every later
                // leaf): from there on this workgroup raises its compute unit's yield counter; the earlier products have whole steps to
                // spare and sleep wherever a workgroup in that state -- or a leaf -- shares their compute unit.  (Every block right of
                // column j starts its product j the moment column j is published: hundreds at a time, two per compute unit, and
                // without this the thirty that matter took 29 us instead of 14.)
                if (m == k && m - j > 1) m = k - 1;            // (the last block column on its own)
                // (only the two blocks under the diagonal: (k+1, k) is what leaf k+1 waits for, (k+2, k) what the block leaf k+2 waits for is
                // updated with as soon as leaf k is through; the blocks further down have a leaf's time to spare)
                const bool last = m == k && row <= k + 2;
                if (last) {
                    if (g.yield && tid == 0) { yslot = cu_yield_slot(g.yield); atomicAdd(yslot, 1); }
                    if (row == k + 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
                }
                // (a tall panel is bound by the products' throughput, not by the leaf chain: nobody sleeps there)
                const int *yb = (row <= k + 2 || g.rows > g.sleep_rows) ? nullptr : g.yield;      // (the two blocks under the diagonal never sleep: their turn comes within a leaf or two)
                unsigned long long bs = 0ull, as = 0ull;
                product<VERIFY, true>(acc, Ar + j * 128, g.lda, Ak + j * 128, g.lda, 8 * (m - j), smem, nullptr, 0, false, &bs, &as, yb);
                if constexpr (VERIFY) {               // both operands were solved and stored by other workgroups
                    unsigned long long wb = 0ull, wa = 0ull;
                    if (tid == 0) for (int jj = j; jj < m; ++jj) { wb += flag_load(g.vhash + VH_ROW + k * 32 + jj); wa += flag_load(g.vhash + VH_ROW + row * 32 + jj); }
                    verify_compare(g, wg_sum(bs, &s_acc), wb);
                    verify_compare(g, wg_sum(as, &s_acc), wa);
                }
                j = m;
            }
        }
        chain_stamp(g, 10, t, row, k);
        if (k == 0 && row <= 2) {                  // (no products: the solve behind leaf 0 starts here)
            if (g.yield && tid == 0) { yslot = cu_yield_slot(g.yield); atomicAdd(yslot, 1); }
            if (row == k + 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
        }
        double4_t xt[8];
        if (k > 0) acc_to_xt(acc, xt, smem);       // (the updated block never goes to memory: only its solve does)
        else trsm_load(xt, Ar + k * 128, g.lda);
        unsigned long long vs = 0ull;
        if (!follow<VERIFY>(g, xt, Ar + k * 128, g.lda, Ak + k * 128, g.lda, g.linv + (long)k * LEAF_DOUBLES, tag + 8ull * k, xcol, smem, &s_i[1], &vs)) {
            if (yslot) atomicAdd(yslot, -1);
            return;
        }
        chain_stamp(g, 3, t, row, k);
        if constexpr (VERIFY) {
            // what arrived of leaf k against what leaf k said it stored (its sum is published with the whole block); then this block's
            // own payload: the solved block as the registers hold it (exactly the bytes stored), published before row_done
            if (!chain_wait(g, F_LEAF, tag + k + 1, &s_i[1])) { if (yslot) atomicAdd(yslot, -1); return; }
            verify_compare(g, wg_sum(vs, &s_acc), flag_load(g.vhash + VH_LEAF + k));
            unsigned long long mine = 0ull;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) mine += (bits_of(xt[tt][0]) + bits_of(xt[tt][1])) + (bits_of(xt[tt][2]) + bits_of(xt[tt][3]));
            mine = wg_sum(mine, &s_acc);
            if (tid == 0) flag_store(g.vhash + VH_ROW + row * 32 + k, mine);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // (the column flag of this row is written by the workgroup of every block of the row in turn: its last value of this block
            // has arrived before the flag goes up that lets the next block's workgroup start)
            flag_store(xcol, tag + 8ull * k + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            flag_store(g.flags + (long)(F_ROW + row) * FL, tag + k + 1);
        }
        chain_stamp(g, 4, t, row, k);
        if (yslot) atomicAdd(yslot, -1);
        return;
    }
    if (t < nsq) {
        // ---- diagonal block `row` of the square: D -= L[row,j] L[row,j]^T behind the solver of its row, block after block, the
        //      block in registers all along (syrk_follow); then its LEAF straight from LDS (no loop around the leaf: inside one the
        //      compiler keeps so much alive across it that it spills hundreds of registers) ----
        const int row = brow;
        double *Ar = g.A + (long)row * 128 * g.lda;
        __builtin_amdgcn_s_setprio(3);
        if (row > 0) {
            const unsigned long long *xcol = g.flags + (long)(F_XCOL + row) * FL;
            double4_t res[5];
            diag_load(res, Ar + row * 128, g.lda);
            int *yslot = nullptr;
            for (int j = 0; j < row; ++j) {
                unsigned long long xs = 0ull;
                // (the last block of the row is the one the leaf waits for: see the solver)
                if (j == row - 1 && g.yield && g.cols_tag == ~0ull && tid == 0) { yslot = cu_yield_slot(g.yield); atomicAdd(yslot, 1); }      // (off: it would hold the counter up for a whole step)
                if (!syrk_follow<VERIFY>(g, res, Ar + j * 128, g.lda, tag + 8ull * j, xcol, smem, &s_i[1], &xs)) { if (yslot) atomicAdd(yslot, -1); return; }
                if constexpr (VERIFY) {               // the block as it arrived against what its solver said it stored
                    if (!chain_wait(g, F_ROW + row, tag + j + 1, &s_i[1])) { if (yslot) atomicAdd(yslot, -1); return; }
                    verify_compare(g, wg_sum(xs, &s_acc), flag_load(g.vhash + VH_ROW + row * 32 + j));
                }
            }
            chain_stamp(g, 5, t, row, row - 1);
            diag_to_tiles(res, smem);
            if (yslot) atomicAdd(yslot, -1);       // (the leaf raises the counter for itself)
        }
        chain_stamp(g, 1, t, row, row);
        LeafArgs la;
        la.A = g.A; la.lda = g.lda; la.linv = g.linv; la.logdet_part = g.logdet; la.info = g.info; la.info_base = g.info_base;
        la.do_factor = g.leaf_factor; la.a_stride = 0; la.linv_stride = 0; la.stamps = g.leaf_stamps; la.tiles_only = g.leaf_tiles; la.yield = g.yield; la.preloaded = row > 0 ? g.leaf_preloaded : 0;
        la.col_flag = g.flags + (long)F_LCOL * FL; la.col_base = tag + 8ull * row;
        const int nv = g.nvalid - 128 * row;
        la.nvalid = nv >= 128 ? 128 : (nv > 0 ? nv : 0);
        if constexpr (!VERIFY) leaf_body<true>(la, Ar + row * 128, g.linv + (long)row * LEAF_DOUBLES, g.logdet + row * 128, g.info_base + 128 * row,
                                               smem, smem + NT * TSZ, tid);
        else {
            unsigned long long ls = 0ull;
            leaf_body<true, true>(la, Ar + row * 128, g.linv + (long)row * LEAF_DOUBLES, g.logdet + row * 128, g.info_base + 128 * row,
                                  smem, smem + NT * TSZ, tid, &ls);
            ls = wg_sum(ls, &s_acc);
            if (tid == 0) flag_store(g.vhash + VH_LEAF + row, ls);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {                            // the last two tile columns went out with the block (the column flag's last value of this block first: the next leaf writes it too)
            flag_store(g.flags + (long)F_LCOL * FL, tag + 8ull * row + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            flag_store(g.flags + (long)F_LEAF * FL, tag + row + 1);
        }
        chain_stamp(g, 2, t, row, row);
        return;
    }

    __builtin_amdgcn_s_setprio(2);
    int *yslot = nullptr;
    const int stride = (int)gridDim.x - nsq;
    double4_t xt[8];
    // (the yield counter goes up AFTER this wait: the update these rows wait for polls that counter on the same compute units)
    if (g.cols_tag && !chain_wait(g, F_COLS, g.cols_tag, &s_i[1])) return;
    if (g.yield_below && g.yield && tid == 0) { yslot = cu_yield_slot(g.yield); atomicAdd(yslot, 1); }
    for (int row = n2 + (t - nsq); row < g.rows; row += stride) {
        // ---- a block row below the square, left-looking ----
        double *Ar = g.A + (long)row * 128 * g.lda;
        for (int k = 0; k < n; ++k) {
            if (k > 0) {
                if (!chain_wait(g, F_ROW + k, tag + k, &s_i[1])) { if (yslot) atomicAdd(yslot, -1); return; }
                chain_stamp(g, 6, t, row, k);
                if constexpr (!VERIFY) product(acc, Ar, g.lda, g.A + (long)k * 128 * g.lda, g.lda, 8 * k, smem, Ar + k * 128, g.lda, false, nullptr, nullptr, g.yield_below ? nullptr : g.yield);
                else {                            // the B operand is row k's solved blocks of steps 0 .. k-1
                    unsigned long long bs = 0ull;
                    product<true>(acc, Ar, g.lda, g.A + (long)k * 128 * g.lda, g.lda, 8 * k, smem, Ar + k * 128, g.lda, false, &bs, nullptr, g.yield_below ? nullptr : g.yield);
                    unsigned long long want = 0ull;
                    if (tid == 0) for (int jj = 0; jj < k; ++jj) want += flag_load(g.vhash + VH_ROW + k * 32 + jj);
                    verify_compare(g, wg_sum(bs, &s_acc), want);
                }
                epilogue<true, false>(acc, Ar + k * 128, g.lda);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            chain_stamp(g, 7, t, row, k);
            trsm_load(xt, Ar + k * 128, g.lda);
            if (!chain_wait(g, F_LEAF, tag + k + 1, &s_i[1])) { if (yslot) atomicAdd(yslot, -1); return; }
            chain_stamp(g, 8, t, row, k);
            if constexpr (!VERIFY) trsm_sub<false>(xt, Ar + k * 128, g.lda, g.A + (long)k * 128 * g.lda + k * 128, g.lda, g.linv + (long)k * LEAF_DOUBLES, smem);
            else {
                unsigned long long vs = 0ull;
                trsm_sub<false, true>(xt, Ar + k * 128, g.lda, g.A + (long)k * 128 * g.lda + k * 128, g.lda, g.linv + (long)k * LEAF_DOUBLES, smem, &vs);
                verify_compare(g, wg_sum(vs, &s_acc), flag_load(g.vhash + VH_LEAF + k));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            chain_stamp(g, 9, t, row, k);
        }
    }
    if (yslot) atomicAdd(yslot, -1);
}

__global__ void chain_cols_ready_kernel(unsigned long long *flag, unsigned long long tag) { flag_store(flag, tag); }

// before a verifying launch: its checksum words <- 0, the last launch's counters added to the running totals behind them
__global__ void chain_verify_begin_kernel(unsigned long long *v) {
    for (int i = threadIdx.x; i < VH_BAD; i += blockDim.x) v[i] = 0ull;
    if (threadIdx.x == 0) { v[VH_WORDS] += v[VH_BAD]; v[VH_WORDS + 1] += v[VH_CHECKS]; v[VH_BAD] = 0ull; v[VH_CHECKS] = 0ull; }
}

// one lane waits up to ~20 ms for *flag to become `tag`: do kernels of two streams really run side by side here?
__global__ void chain_probe_wait_kernel(const unsigned long long *flag, unsigned long long tag, int *seen) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000000ull) {          // 100 MHz
        if (flag_load(flag) == tag) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    *seen = ok;
}

}  // namespace

// A kernel that waits in memory for a kernel of ANOTHER stream needs the two to run concurrently.  Profilers that collect hardware
// counters, and serialising debug modes of the runtime, run one kernel at a time: the wait would only end by its timeout.  Probed
// once per handle (a waiting kernel on the chain stream, the kernel that raises the flag on the main stream, ~0.1 ms): without
// concurrency the split column update (potrf_driver, `cols_split`) stays off.
int chain_streams_concurrent(fvgp_handle *h) {
    if (h->streams_concurrent >= 0) return h->streams_concurrent;
    h->streams_concurrent = 0;
    if (fvgp_ensure_side(h)) return 0;
    if (!h->chain_flags) {
        if (hipMalloc((void **)&h->chain_flags, (size_t)FLAG_LINES * FL * sizeof(unsigned long long)) != hipSuccess) return 0;
        if (hipMemset(h->chain_flags, 0, (size_t)FLAG_LINES * FL * sizeof(unsigned long long)) != hipSuccess) return 0;
        h->chain_tag = 0; h->chain_tick = 0;
    }
    h->chain_tag += 512;
    int *seen = h->dinfo + 2;
    unsigned long long *flag = h->chain_flags + F_COLS * FL;
    hipEvent_t e0 = nullptr;
    if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess) return 0;
    // both streams first wait for the handle's stream (whichever it is now), then: side waits in memory, main raises the flag
    (void)hipEventRecord(e0, h->stream);
    (void)hipStreamWaitEvent(h->side, e0, 0);
    hipLaunchKernelGGL(chain_probe_wait_kernel, dim3(1), dim3(1), 0, h->side, flag, h->chain_tag, seen);
    hipLaunchKernelGGL(chain_cols_ready_kernel, dim3(1), dim3(1), 0, h->stream, flag, h->chain_tag);
    int host_seen = 0;
    if (hipStreamSynchronize(h->side) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess &&
        hipMemcpy(&host_seen, seen, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) h->streams_concurrent = host_seen ? 1 : 0;
    (void)hipEventDestroy(e0);
    return h->streams_concurrent;
}

// behind the update of a panel's rows below its square: the resident kernel's block rows below the square may read them now
// (a kernel boundary lies between that update and this store: its results are in memory, every L2 has been invalidated since)
int launch_chain_cols_ready(fvgp_handle *h, unsigned long long tag) {
    hipLaunchKernelGGL(chain_cols_ready_kernel, dim3(1), dim3(1), 0, h->stream, h->chain_flags + F_COLS * FL, tag);
    HIPCHK(hipGetLastError());
    return 0;
}

// One panel [J0, J0 + 128 n) of the padded np x np matrix A, every row from J0 down: factor, solve, update -- one launch.
// cols_tag_out != nullptr: the block rows below the square wait for launch_chain_cols_ready(*cols_tag_out) on another stream.
int launch_panel_chain(fvgp_handle *h, double *A, int64_t n_valid, int64_t np, int64_t lda, int64_t J0, int64_t Jend, unsigned long long *cols_tag_out) {
    const int64_t w = Jend - J0;
    if (w <= 0 || w % TILE || J0 % TILE || np % TILE || Jend > np || w / TILE > FVGP_CHAIN_MAX_BLOCKS) { fvgp_set_error("panel chain: bad panel"); return -5; }
    if (lda >= (1L << 21) || (lda & 1) || ((uintptr_t)A & 15)) { fvgp_set_error("panel chain: leading dimension / alignment"); return -4; }
    if (!h->chain_flags) {
        HIPCHK(hipMalloc((void **)&h->chain_flags, (size_t)FLAG_LINES * FL * sizeof(unsigned long long)));
        HIPCHK(hipMemset(h->chain_flags, 0, (size_t)FLAG_LINES * FL * sizeof(unsigned long long)));
        h->chain_tag = 0; h->chain_tick = 0;
    }
    ChainArgs g;
    g.A = A + J0 * lda + J0; g.lda = (long)lda; g.n = (int)(w / TILE); g.rows = (int)((np - J0) / TILE);
    g.nvalid = (int)(n_valid - J0 > 0 ? n_valid - J0 : 0);
    g.linv = h->linv + (J0 / TILE) * LEAF_DOUBLES; g.logdet = h->logdet_parts + J0;
    g.info = h->dinfo; g.info_base = (int)J0;
    g.flags = h->chain_flags;
    h->chain_tag += 512;                  // (a launch uses tag + 1 .. tag + 8 n <= tag + 256 for its flags)
    g.tag0 = h->chain_tag; g.tick0 = h->chain_tick;
    g.cols_tag = cols_tag_out ? h->chain_tag : 0;
    if (cols_tag_out) *cols_tag_out = h->chain_tag;
    g.yield = h->leaf_yield ? h->cu_yield : nullptr;
    g.stamps = h->chain_stamps; g.seq = h->chain_seq++; g.leaf_stamps = h->leaf_stamps; g.leaf_factor = 1; g.leaf_tiles = 1; g.leaf_preloaded = 1; g.yield_below = h->chain_yield >= 2;
    // one ticket per block of the first n2 block rows (the square, and as many rows below it as keep that under ~480 workgroups: short
    // panels have a workgroup per block), then one per block row (dealt round-robin beyond 480 workgroups in all, at least 128 of these)
    // (beside a trailing update every resident workgroup holds a slot of the update: only the square -- the critical path -- has a
    // workgroup per block there, most of which wait most of the time; alone on the chip waiting costs nothing)
    int n2 = h->chain_alone ? g.rows : g.n;
    if (n2 > g.rows) n2 = g.rows;
    if (n2 > VH_ROWS) n2 = VH_ROWS;
    g.n2 = n2; g.sleep_rows = h->chain_sleep_rows;
    const int nsq = g.n * n2 - g.n * (g.n - 1) / 2;
    int below = g.rows - n2;
    const int room = 480 - nsq > 128 ? 480 - nsq : 128;
    if (below > room) below = room;
    const int grid = nsq + below;
    h->chain_tick += (unsigned long long)grid;
    // option "chain_ahead" (default 0 = plain column order; measured: no gain): alone on the chip (>= 256 slots) the critical tasks are
    // dealt that many columns in front; never beside an update or with rows below the blocks
    g.ahead = (h->chain_alone == 1 && n2 == g.rows) ? h->chain_ahead : 0;
    g.vhash = nullptr;
    if (h->chain_verify) {
        // per launch: VH_WORDS checksum words, zeroed on the launch's stream; the counters of all launches add up in the handle's
        // pair of words behind them (read by fvgp_hip_chain_verify_counts)
        if (!h->chain_vhash) {
            HIPCHK(hipMalloc((void **)&h->chain_vhash, (size_t)(VH_WORDS + 2) * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(h->chain_vhash, 0, (size_t)(VH_WORDS + 2) * sizeof(unsigned long long), h->stream));
        }
        g.vhash = h->chain_vhash;
        hipLaunchKernelGGL(chain_verify_begin_kernel, dim3(1), dim3(256), 0, h->stream, h->chain_vhash);
        hipLaunchKernelGGL(chain_kernel<true>, dim3((unsigned)grid), dim3(512), 0, h->stream, g);
    } else {
        // a short panel alone on the chip is bound by its chain of leaves: with ONE workgroup per compute unit (16 KB of LDS nobody uses
        // take the second one's room) a leaf, or the solve behind it, never shares its SIMDs with another block's products
        const unsigned pad_lds = (h->chain_alone == 1 && g.rows <= h->chain_single_rows) ? 16384u : 0u;      // (beside an update the padded workgroup would not fit next to an update workgroup)
        hipLaunchKernelGGL(chain_kernel<false>, dim3((unsigned)grid), dim3(512), pad_lds, h->stream, g);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// chain_verify: {mismatches, comparisons} over every resident panel kernel launched on this handle since the last call
int chain_verify_counts(fvgp_handle *h, unsigned long long *out2) {
    out2[0] = out2[1] = 0;
    if (!h->chain_vhash) return 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    unsigned long long w[4];
    HIPCHK(hipMemcpy(w, h->chain_vhash + VH_BAD, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    out2[0] = w[0] + w[2]; out2[1] = w[1] + w[3];
    HIPCHK(hipMemset(h->chain_vhash + VH_BAD, 0, 4 * sizeof(unsigned long long)));
    return 0;
}

// host-only replay of the panel kernel's ticket -> task map for a panel of n block columns whose first n2 block rows have a workgroup per
// block, the critical tasks dealt `ahead` columns in front (chain_ticket_role): out3 = {kind, block row, block column}
extern "C" int fvgp_hip_debug_chain_ticket(int n, int n2, int ahead, int ticket, int *out3) {
    if (n < 1 || n > FVGP_CHAIN_MAX_BLOCKS || n2 < n || ahead < 0 || ticket < 0 || !out3) return -1;
    chain_ticket_role(n, n2, ahead, ticket, &out3[0], &out3[1], &out3[2]);
    return 0;
}
