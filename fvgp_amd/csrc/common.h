// Shared host-side declarations of libfvgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/fvgp_hip.h"

constexpr int TILE = FVGP_TILE;        // 128: tile edge of every kernel and the leaf Cholesky block
constexpr int LEAF_DOUBLES = TILE * TILE;
constexpr int CU_YIELD_STRIDE = 32;    // ints between two compute units' yield counters (one 128-byte line each)
constexpr int CU_YIELD_KEYS = 8 * 256;   // XCC id (3 bits) << 8 | HW_ID[15:8] (CU, SH, SE)
#if defined(__HIPCC__)
// a wave-uniform pointer moved into SGPRs (readfirstlane returns int: widen each half as UNSIGNED)
__device__ __forceinline__ const double *uniform_ptr(const double *p) {
    const uintptr_t v = (uintptr_t)p;
    const uintptr_t lo = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffu));
    const uintptr_t hi = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const double *>(lo | (hi << 32));
}
// this compute unit's yield counter (fvgp_handle::cu_yield): raised by the latency-bound kernels of the panel chain while one
// of their workgroups runs here, polled by the trailing update's waves (gemm.hip, YIELD)
__device__ __forceinline__ int *cu_yield_slot(int *base) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 8)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    return base + ((xcc << 8 | hw) * CU_YIELD_STRIDE);
}
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// XCD-balanced block -> tile tables of the GEMM launches, one per launch shape (gemm.hip)
struct TileTabKey {
    int tm, tn, lower, ls, lo;
    bool operator<(const TileTabKey &o) const {
        if (tm != o.tm) return tm < o.tm;
        if (tn != o.tn) return tn < o.tn;
        if (lower != o.lower) return lower < o.lower;
        if (ls != o.ls) return ls < o.ls;
        return lo < o.lo;
    }
};
struct TileTab { int *dev; long grid; int *host; };      // host: pinned staging copy (the upload reads it from a copy stream)

struct fvgp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    // inverses of the 128x128 diagonal blocks of the last factor (row-major, upper = 0)
    double *linv = nullptr;
    size_t linv_blocks = 0;
    const double *linv_L = nullptr;   // which factor they belong to
    int64_t linv_n = 0, linv_ld = 0;
    // inverses of the 1024 x 1024 diagonal blocks of the same factor (block J in rows [1024 J, ..) of an np x 1024 array),
    // built on demand from the 128-block inverses by doubling (ensure_winv) for the many-right-hand-side forward
    // substitution of the posterior; dropped whenever the 128-block inverses change owner
    double *winv = nullptr;
    size_t winv_cap = 0;
    bool winv_ok = false;
    int64_t winv_w = 1024;            // width of winv's layout (row stride)
    int64_t winv_level = 1024;        // size of the inverted diagonal blocks it holds so far (<= winv_w)
    int64_t posterior_block = 2048;   // block width of the many-point posterior's sweep up to 1024 points (1024 / 2048); 1024 beyond
    int64_t leaf_tiles_rows = 4096;   //   ... while at most this many rows lie below the block
    int panel_recursive = 1;          // option: panels are factored by recursive halving (0: 128-column steps inside `inner_block` sub-panels)
    int leaf_tiles = 1;               // option: the leaf leaves the inverses of its 16x16 diagonal tiles only, the chain's TRSM substitutes
    int k128_kernels = 1;             // option: K = 128 products of the panel chain fetch their operands in one stage
    int block_inverses = 1;           // option: 0 = the posterior substitution walks the 128-blocks instead
    int potri_kminor = 1;             // option: POTRI on the (M,K) x (N,K) products only (0: the round-2 schedule with (K,N) / (K,M) operands)
    // per-leaf sum(log L_ii), device
    double *logdet_parts = nullptr;
    size_t logdet_cap = 0;
    // small device scratch: reductions, info flag
    double *red = nullptr;            // RED_SLOTS doubles
    int *dinfo = nullptr;
    double *hpin = nullptr;           // pinned host mirror (RED_SLOTS doubles)
    // scratch for the vector solves (padded_n x 8)
    double *vec = nullptr;
    size_t vec_cap = 0;
    // options
    int64_t outer_block = 1024;
    int64_t outer_block_big = 2048, big_threshold = 24576;   // wider panels while the trailing matrix is large
    int profile = 0;
    int64_t inner_block = 512;        // sub-panel width inside panels wider than this (0 = off)
    int64_t small_tile_max_update = 512;   // trailing updates of at most this many 128-tiles also run on 64-tiles
    int64_t small_tile_max = 160;     // (M,K) x (N,K) products of at most this many 128-tiles and K <= 512 run on 64-tiles
    unsigned long long *chain_stamps = nullptr; int chain_seq = 0;   // diagnostics (option "chain_stamps"): trsm_tiles workgroup start / end times
    unsigned long *leaf_stamps = nullptr;   // diagnostics (option "leaf_stamps" = device pointer): phase timestamps of the leaf kernel
    std::map<TileTabKey, TileTab> tile_tabs;   // device-resident, freed with the handle
    int tile_tables = 1;              // plain launches of the 128-tile kernels take the XCD-balanced tile table instead of the formula map
    int n_cus = 256;
    // cooperative yield: one counter per compute unit (key = XCC id << 8 | HW_ID[15:8], a line of its own each).  A leaf that
    // runs under look-ahead raises its CU's counter; the trailing-update waves of that CU poll it once per K step with a scalar
    // load and sleep while it is up (fp64 MFMA and the vector ALU share a pipe: beside an MFMA stream every dependent instruction
    // of the latency-bound leaf waits for a 64-cycle MFMA -- 3.5 to 6.7 times the standalone time).  Option "leaf_yield".
    int *cu_yield = nullptr; int leaf_yield = 1, chain_yield = 1;      // chain_yield 2: the resident panel kernel's block rows below the square raise it too
    // backward sweep in one launch (solve.hip, bwd_sweep_kernel): granules of {value, tag}, the launch counter the tags come from,
    // the column ticket; option "bwd_sweep"
    double *sweep_gran = nullptr; size_t sweep_gran_cap = 0; unsigned long long sweep_tag = 0; int *sweep_ticket = nullptr; int bwd_sweep = 1, fwd_sweep = 1;
    // the panel chain as one resident kernel per panel (chain.hip): flag words, the launch tag and ticket bases; option "panel_chain"
    unsigned long long *chain_flags = nullptr, chain_tag = 0, chain_tick = 0; int panel_chain = 1;
    int streams_concurrent = -1;      // -1: not probed yet (chain.hip, chain_streams_concurrent)
    // the default schedule (potrf_driver): panels of `wide_block` columns (`wide_block_big` while more than `wide_threshold` rows remain), each
    // ONE resident kernel alone on the chip + one trailing update; a panel over >= wide_inner_rows rows in sub-panels of wide_inner columns
    int chain_wide = 1; int64_t wide_block = 4096, wide_block_big = 4096, wide_threshold = 1 << 30, wide_inner = 2048, wide_inner_rows = 16384;
    int chain_alone = 1;              // set per factorisation: 1 no update runs beside the panel kernels, 0 look-ahead, 2 the row-sharded driver (a workgroup per block, not alone)
    int chain_ahead = 0;              // measurement only: alone on the chip, block columns the critical tasks (diagonal block + the two blocks under it) are dealt in front of the bulk tasks (chain.hip; no gain measured)
    int chain_sleep_rows = 96;        // panels of at most this many block rows: early products yield their compute unit to the critical blocks
    int chain_single_rows = 80;       // panels of at most this many block rows run one workgroup per compute unit (alone only; 96 until round 6: N=12k -2.8 %, N=16k -1.7 %, N=8k -1 %)
    int chain_verify = 0; unsigned long long *chain_vhash = nullptr;   // option "chain_verify": payload checksums on every hand-off of the resident panel kernel (chain.hip, VH_*)
    int cols_split = 1; int64_t cols_split_rows = 8192;   // (potrf_driver: the next panel's square is updated first, the rows below it beside its chain, while at most this many rows remain)
    int64_t panel_chain_min = 4096;   // ... for panels with at least this many rows from their first column down (below that the chain runs alone on the chip and the three launches per step are as fast)
    // look-ahead (the next panel factored on a second stream beside the trailing update) from this many (padded) rows on.  Off by default
    // since the resident panel kernel has a workgroup per block (chain.hip): 4096-wide panels factored ALONE on the whole chip, each
    // followed by one K = 4096 update with nothing beside it, beat look-ahead at every size measured (N=2000 0.95 -> 0.75 ms,
    // 8000 5.7 -> 5.1, 20000 46.9 -> 45.6, 50000 620 -> 595 on the same box); `lookahead_min` = 4608 restores the old schedule
    int64_t lookahead_min = (int64_t)1 << 40;
    int posterior_halves = 1;         // posterior covariance at >= 512 points: two halves of the points side by side on two streams (api.hip)
    int64_t outer_block_small = 512, small_threshold = 12288;   // panel width for the last `small_threshold` rows (potrf_driver)
    int lookahead = 1;
    double *tr_ws = nullptr; size_t tr_ws_cap = 0;    // trsm_lower with few columns: the right-hand sides transposed
    hipStream_t side = nullptr;       // high-priority stream for the look-ahead panel
    hipEvent_t ev_panel = nullptr, ev_cols = nullptr;
    // profile of the last potrf
    std::vector<hipEvent_t> ev;
    std::vector<double> ev_flops, ev_bytes;
    double prof_launches = 0, prof_ms = 0, prof_flops = 0, prof_total_ms = 0, prof_bytes = 0;
    double prof_host_enqueue_ms = 0;   // row-sharded evaluation: host time to enqueue one evaluation (no synchronisation inside)
    double prof_kmat_ms = 0, prof_kmat_bytes = 0, prof_tail_ms = 0;   // fused evaluation: assembly, everything after the factorisation
    hipEvent_t ev_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    // row-sharded trailing updates timed since the last get_profile (option "profile")
    hipStream_t copy_stream = nullptr;   // uploads of tile tables: the host waits for this stream only, never for the compute streams
    size_t tile_tab_bytes = 0;           // device bytes held by tile_tabs (capped, gemm.hip)
    std::vector<hipEvent_t> rs_ev;
    std::vector<double> rs_flops;
    size_t rs_used = 0;
    // row-sharded evaluation: the collectives (RCCL or the caller's), this rank's place, timings of the calls (option "profile")
    fvgp_collectives coll{nullptr, nullptr, nullptr};
    int coll_rank = 0, coll_nranks = 1;
    void *rccl_lib = nullptr, *rccl_comm = nullptr;
    void *ipc_comm = nullptr;          // direct collectives over peer mappings (ipc.hip), or nullptr
    struct CollRec { int kind; double bytes; hipEvent_t e0, e1; };
    std::vector<CollRec> coll_rec;
    std::vector<hipEvent_t> coll_ev_pool;
};
constexpr int RED_SLOTS = 4096;

void fvgp_set_error(const std::string &s);
int fvgp_hip_fail(hipError_t e, const char *what, int line);

#define HIPCHK(call)                                                  \
    do {                                                              \
        hipError_t e__ = (call);                                      \
        if (e__ != hipSuccess) return fvgp_hip_fail(e__, #call, __LINE__); \
    } while (0)

static inline int64_t pad128(int64_t n) { return (n + TILE - 1) / TILE * TILE; }

// ---- launches implemented in the .hip units ------------------------------------------------
struct GemmDesc {
    int a_kmajor, b_nmajor, lower;    // lower: 0 all tiles, 1 tile row >= tile col, 2 tile col <= tile row * lower_scale + lower_off
    int lower_scale = 1, lower_off = 0;
    int role = 0;                     // 1 = trailing update of the Cholesky (own kernel symbol for profilers)
    int64_t M, N, K;
    double alpha, beta;
    const double *A; int64_t lda;
    const double *B; int64_t ldb;
    double *C; int64_t ldc;
    // per-tile K range: [kb0 + kbi*ti + kbj*tj, ke0 + kei*ti + kej*tj) clamped to [0,K], in elements
    int64_t kb0 = 0, kbi = 0, kbj = 0, ke0 = -1, kei = 0, kej = 0;
    // B (b_nmajor = 0 only) as an all-gather leaves it: `bc_ranks` chunks of `bc_blocks` 128-row blocks, chunk q
    // holding the blocks q, q + bc_ranks, ...; output tile column tj reads block tj + bc_off of that cyclic order
    int bc_ranks = 1, bc_blocks = 0, bc_off = 0;
    int rev_m = 0;                    // walk the tile rows from the last to the first (per-tile K grows with ti: longest first)
    // split-K for products with few output tiles and a long K (S -= V^T V of the posterior: 64 tiles, K = N): `split`
    // workgroups per tile, each over K/split, partial tiles into split_ws (split x M x N doubles), then one fixed-order
    // reduction into C -- the result does not depend on scheduling.  Plain K range only.
    int split = 1;
    int split_tri = 0;                // B (N, K) is lower triangular: tile column tj only has K < 128 (tj + 1); slices beyond are skipped
    double *split_ws = nullptr;
    double *split_out = nullptr; int64_t split_ldo = 0;   // the reduced result beta C + sum goes here instead of over C
    // strided batch of equal problems: problem (y, z), y < batch_y, z < batch_z, takes A + y a_by + z a_bz, B + .., C + ..
    // (elements).  Excludes split-K.
    int batch_y = 1, batch_z = 1;
    int64_t a_by = 0, a_bz = 0, b_by = 0, b_bz = 0, c_by = 0, c_bz = 0;
};
int launch_gemm(fvgp_handle *h, const GemmDesc &g);
bool gemm_takes_small_tiles(const fvgp_handle *h, const GemmDesc &g);   // the launch runs gemm_f64_small_kernel, not gemm_f64_kernel
void gemm_release_tables(fvgp_handle *h);
long gemm_debug_tile_table(int tiles_m, int tiles_n, int lower, int ls, int lo, int *out, long cap);
long gemm_debug_tile_map(int tiles_m, int tiles_n, int lower, int ls, int lo, int *out_ti, int *out_tj, long cap);

struct KmatDesc {
    int kind;                 // 0 rbf, 1 matern 3/2, 2 matern 5/2
    const double *x1; int64_t n1;
    const double *x2; int64_t n2;
    int d;
    double sig;
    double invl[FVGP_MAX_DIM];
    const double *vdiag;
    double *K; int64_t ldk;
    int uplo, pad;
};
int launch_kmat(fvgp_handle *h, const KmatDesc &k);
int kmat_desc_from_theta(int kernel_id, int d, const double *theta, int ntheta, KmatDesc *out);

struct GradDesc {
    KmatDesc k;               // x1 == x2 == x, n1 == n2 == n
    int kernel_id;
    int ntheta;
    const double *W; int64_t ldw;   // lower triangle of KV^-1
    const double *b; int64_t ldb;   // KVinvY column (stride ldb)
    double *partial;                // device, nblocks x ntheta
    int64_t col0 = 0, ncols = 0;    // ncols > 0: W is the slab of columns [col0, col0 + ncols) (col0 % 128 == 0), its column 0 = matrix column col0
};
int launch_grad_trace(fvgp_handle *h, const GradDesc &g, int *nblocks_out);

int launch_panel_chain(fvgp_handle *h, double *A, int64_t n_valid, int64_t np, int64_t lda, int64_t J0, int64_t Jend, unsigned long long *cols_tag_out = nullptr);
int launch_chain_cols_ready(fvgp_handle *h, unsigned long long tag);
int chain_streams_concurrent(fvgp_handle *h);
int chain_verify_counts(fvgp_handle *h, unsigned long long *out2);
int launch_leaf(fvgp_handle *h, double *A, int64_t lda, double *linv, double *logdet_part, int info_base, int do_factor, int nvalid,
                int tiles_only = 0);
// X = A inv(L)^T in place for `rows` (a multiple of 32) rows of 128 columns, by substitution over the eight 16-column tiles of
// the 128 x 128 lower block L with the inverses of its diagonal tiles (`dinv`, as launch_leaf(tiles_only) leaves them)
int launch_trsm_tiles(fvgp_handle *h, double *A, int64_t lda, int64_t rows, const double *L, int64_t ldl, const double *dinv);
int launch_leaf_inverse_batched(fvgp_handle *h, const double *L, int64_t ldl, int64_t nblk, double *linv);

int launch_fwd_step(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv_k,
                    double *B, int64_t ldb, double *Y, int c);
int launch_neg_log_sum(fvgp_handle *h, const double *v, int64_t n, double *out_dev);
int launch_loglik_tail(fvgp_handle *h, const double *v, int64_t nlog, const double *A, int64_t lda, int64_t n, int ncol, double *out2_dev,
                       double *vec, int C, int64_t np, double *alpha);
int launch_bwd_sweep(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, const double *linv, const double *Yres, double *X, int64_t ldx, int c);
int launch_fwd_sweep(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, const double *linv, const double *B, int64_t ldb, double *Y);
int launch_bwd_step(fvgp_handle *h, const double *L, int64_t ldl, int64_t np, int64_t k0, const double *linv_k,
                    double *Yres, double *X, int64_t ldx, int c);
int launch_diag_logsum(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *out_dev);
int launch_sum(fvgp_handle *h, const double *v, int64_t n, double *out_dev);
int launch_dot_rows(fvgp_handle *h, const double *a, int64_t lda, const double *b, int64_t ldb, int64_t n, int c, double *out_dev);
int launch_add_lower(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t n, double alpha);
int launch_coldot(fvgp_handle *h, const double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t P, double *out);
int launch_add_matrix(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double alpha);
int launch_pad_identity(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda);
int launch_rhs_rows(fvgp_handle *h, double *A, int64_t n, int64_t lda, const double *ymean, int ncol, const double *vdiag);
int launch_rows_to_vec(fvgp_handle *h, const double *A, int64_t lda, int64_t row0, int nrows, double *vec, int C, int64_t np);
int launch_rowsumsq(fvgp_handle *h, const double *A, int64_t lda, int64_t row0, int nrows, int64_t ncols, double *out_dev);
int launch_copy_cols(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t rows, int64_t cols,
                     int64_t rows_pad, int64_t cols_pad);
int launch_symmetrize(fvgp_handle *h, double *A, int64_t n, int64_t lda);
int launch_colsumsq(fvgp_handle *h, const double *V, int64_t rows, int64_t ldv, int64_t P, double base, double *out, double sign);   // out = base - sign * sum_i V[i][p]^2
// row-wise twins for the transposed cross-covariance block KT (P x n, leading dimension ldk):
// out[p][c] = sum_n KT[p][n] alpha[n][c]   and   out[p] = base - sum_n KT[p][n]^2
int launch_rows_dot(fvgp_handle *h, const double *KT, int64_t ldk, const double *alpha, int64_t lda, int ncol, int64_t n, int64_t P,
                    double *out, int64_t ldo);
int launch_rows_sumsq_base(fvgp_handle *h, const double *KT, int64_t ldk, int64_t n, int64_t P, double base, double *out);
int launch_splitk_reduce(fvgp_handle *h, const double *ws, int split, int64_t M, int64_t N, int lower, const double *C, int64_t ldc, double beta,
                         double *out, int64_t ldo, int64_t tri_ksplit = 0);
int launch_winv_seed(fvgp_handle *h, const double *linv, int64_t nblk, double *W, int64_t w = 1024);
int launch_mfma_selftest(fvgp_handle *h, const double *A, const double *B, double *D);
int launch_mfma_peak(fvgp_handle *h, double *out, int blocks, int iters);
int launch_copy_lower_tiles(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t np);
int launch_transpose(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t rows, int64_t cols);
int launch_transpose_lower_tiles(fvgp_handle *h, const double *src, int64_t lds, double *dst, int64_t ldd, int64_t np);
int launch_trace_dot(fvgp_handle *h, const double *W, int64_t ldw, const double *D, int64_t ldd, const double *b, int64_t ldb, int64_t n,
                     double *partial, int *nblocks);

int ensure_linv(fvgp_handle *h, const double *L, int64_t n, int64_t ldl);
int ensure_scratch(fvgp_handle *h, int64_t np);
int fvgp_ensure_side(fvgp_handle *h);
void fvgp_ipc_destroy(fvgp_handle *h);
int fvgp_ipc_check(fvgp_handle *h);      // 2200 once a poll of the direct collectives has given up (ask AFTER synchronising), else 0
int fvgp_read_back(fvgp_handle *h, const double *dev, double *host, int count);
