/*
 * fvgp_hip.h -- flat C ABI of the MI355X-native exact-GP engine (libfvgp_hip.so).
 *
 * The library replaces the numpy/scipy calls on fvGP's dense hot path.  Each entry point
 * names the reference interface it stands in for (paths relative to the lbl-camera/fvGP
 * root).  The reference is pure Python, so the binding a maintainer adds is a ctypes stub
 * (INTEGRATION.md); fvgp_amd/_lib.py is that stub.
 *
 * Conventions
 *   - every matrix is fp64, ROW-MAJOR (numpy / torch default) with an explicit leading
 *     dimension in elements; device pointers unless the parameter says "host";
 *   - square factor / covariance buffers are PADDED: rows and leading dimension are
 *     fvgp_hip_padded_dim(n) (a multiple of 128); the library owns the contents of the
 *     padding (identity on the diagonal, zero elsewhere) so every kernel runs on whole
 *     128x128 tiles;
 *   - only the LOWER triangle (i >= j) of a symmetric input / Cholesky output is defined,
 *     exactly like scipy.linalg.cho_factor(lower=True) (fvgp/gp_lin_alg.py:245);
 *   - the caller (torch tensors on the Python side) owns every buffer; the handle owns
 *     only its scratch (diagonal-block inverses, reduction slots);
 *   - return value: 0 ok; -k = argument k (1-based) is invalid (LAPACK style);
 *     >= 1000 = HIP runtime failure (text in fvgp_hip_last_error_string);
 *     "info" out-parameters follow dpotrf: 0, or the order of the first leading minor
 *     that is not positive definite.
 *   - a handle is bound to one device and one stream and is not thread-safe; distinct
 *     handles are independent.  Calls that return host scalars synchronise the stream;
 *     all others are asynchronous on it.
 */
#ifndef FVGP_HIP_H
#define FVGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fvgp_handle fvgp_handle;

/* stationary kernels of fvgp/kernels.py; theta = [signal variance, l_1..l_d] (ARD)
 * or [signal variance, l] (isotropic) */
enum fvgp_kernel_id {
    FVGP_KERNEL_RBF_ARD = 0,      /* hps[0]*squared_exponential_kernel(get_anisotropic_distance_matrix(..,hps[1:]),1)  kernels.py:16-33,461-481 */
    FVGP_KERNEL_MATERN32_ARD = 1, /* GPprior._default_kernel                                                            gp_prior.py:376-400 */
    FVGP_KERNEL_MATERN52_ARD = 2, /* gp_bo._surrogate_kernel                                                            gp_bo.py:115-126 */
    FVGP_KERNEL_RBF_ISO = 3,      /* hps[0]*squared_exponential_kernel(get_distance_matrix(x1,x2),hps[1])               kernels.py:440-458 */
    FVGP_KERNEL_MATERN32_ISO = 4, /* hps[0]*matern_kernel_diff1(get_distance_matrix, hps[1])                            kernels.py:98-118 */
    FVGP_KERNEL_MATERN52_ISO = 5  /* hps[0]*matern_kernel_diff2(get_distance_matrix, hps[1])                            kernels.py:166-188 */
};

enum fvgp_uplo { FVGP_FULL = 0, FVGP_LOWER = 1 };

#define FVGP_TILE 128
#define FVGP_MAX_DIM 16     /* input dimension limit of the assembly kernels */
#define FVGP_MAX_RHS_VEC 8  /* potrs switches from the GEMV path to the GEMM path above this */
#define FVGP_CHAIN_MAX_BLOCKS 32   /* widest panel (in 128-column blocks) the resident panel kernel takes; wider ones use the launch-per-step chain */

int fvgp_hip_version(void);
const char *fvgp_hip_last_error_string(void);
int64_t fvgp_hip_padded_dim(int64_t n);
/* rows AND columns of the square scratch fvgp_hip_loglik_rows wants for n points and ncol columns of y: padded_dim(n) while that
 * leaves ncol padding rows for the appended (y-m)^T, else 128 more (n a multiple of 128: gp_kv.py:589-593's forward solve then
 * rides in the factorisation for every n).  A scratch of only padded_dim(n) rows is accepted: the forward solve is then a sweep of
 * its own. */
int64_t fvgp_hip_loglik_dim(int64_t n, int ncol);
/* device bytes a handle allocates by itself for problems of n points (npred prediction points, 0 = none);
 * every N x N buffer is the caller's (gp_kv.py keeps Chol_factor / KVinvY as attributes the same way) */
int64_t fvgp_hip_workspace_bytes(int64_t n, int64_t npred);

/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL; it must outlive the handle
 * (fvgp_hip_destroy synchronises it) */
int fvgp_hip_create(fvgp_handle **out, int device, void *stream);
int fvgp_hip_destroy(fvgp_handle *h);
int fvgp_hip_sync(fvgp_handle *h);
/* a stream restricted to the compute units whose bits are set in cu_mask (mask_words x 32 bits), or an
 * ordinary non-blocking stream (high_priority 0/1) when cu_mask is NULL.  The row-sharded driver gives
 * the panel chain a few CUs of its own so its small kernels never queue behind the trailing update. */
int fvgp_hip_stream_create(void **out_stream, int device, int high_priority, const uint32_t *cu_mask, int mask_words);
int fvgp_hip_stream_destroy(void *stream);
/* Options.  Three keys are the library's interface; none changes a result except where noted "order": another, equally valid order of
 * the same sums (LAPACK accuracy either way: dpotrf behind gp_lin_alg.py:245 has no fixed order either).
 *   "schedule" ("order"): how a matrix is factored.
 *       0 = wide (default): panels of 4096 columns, each ONE resident kernel (csrc/chain.hip, a workgroup per 128 x 128 block) ALONE on
 *           the chip, followed by ONE trailing update with K = the panel's width; a panel over at least 16384 rows goes in sub-panels of
 *           2048 columns with the update kernel bringing the rest of the panel up to date in between;
 *       1 = lookahead: panels of 2048 / 1024 / 512 columns, the next panel factored on a high-priority side stream under the trailing
 *           update (the default until round 5; the row-sharded driver still runs its stacked panels this way);
 *       2 = narrow: the same panels, three launches per 128 columns instead of the resident kernel (panels wider than 4096 columns
 *           always take this form).
 *   "profile" (0/1): time the trailing-update launches with HIP events -> fvgp_hip_get_profile.
 *   "chain_verify" (0/1): checksummed hand-offs of the resident panel kernel, see fvgp_hip_chain_verify_counts.
 * Everything below is a MEASUREMENT knob underneath one of the schedules -- kept because each has a test
 * (tests/test_gpu_primitives.py) and a measured A/B behind its default; not an interface, no promise that a key
 * survives a round:
 *   wide: "chain_wide" (1), "wide_block" = 4096 ("wide_block_big" while more than "wide_threshold" rows remain), "wide_inner" = 2048 and
 *       "wide_inner_rows" = 16384 (the sub-panels);
 *   lookahead / narrow: "lookahead_min" (2^40 = off; padded size from which look-ahead runs, 4608 under schedule 1), "lookahead" (1),
 *       "outer_block" (1024; "outer_block_big" = 2048 while more than "big_threshold" = 24576 rows remain, "outer_block_small" = 512
 *       for the last "small_threshold" = 12288; any multiple of 128),
 *   panel chain: "panel_chain" (1: one resident kernel per panel -- in the look-ahead schedule for panels with at least
 *       "panel_chain_min" = 4096 rows below their first column -- and for the row-sharded driver's stacked panel; 0: three launches
 *       per 128 columns ("inner_block" / "panel_recursive": how those split a panel); 2: in the row-sharded driver a workgroup per
 *       block ROW below the square instead of per block) ("order"), "chain_sleep_rows" (96: in panels of at most this many block rows
 *       a block's early products yield their compute unit to a leaf or to the block the next leaf waits for), "chain_single_rows"
 *       (80: panels of at most this many block rows run one workgroup per compute unit), "chain_ahead" (0 = plain column order: alone on
 *       the chip the diagonal block and the two blocks under it of the next block columns are started this many columns ahead of the
 *       other blocks; the leaves of a tall panel end earlier, the launch does not: profiles/r06_chain_ahead_ab.txt),
 *       "cols_split" (look-ahead schedule; 1: while at most "cols_split_rows" = 8192 rows remain only the next panel's square is
 *       brought up to date before its resident kernel starts; the rows below follow on the main stream and the kernel's block rows
 *       wait for a flag in memory),
 *       "leaf_tiles" / "leaf_tiles_rows" / "k128_kernels" / "small_tile_max" / "small_tile_max_update" (kernels of the three-launch
 *       chain) ("order"), "leaf_yield" / "chain_yield" (1: the trailing update's waves sleep while a workgroup of the chain shares
 *       their compute unit; chain_yield 2: also for the resident kernel's rows below the square); "tile_tables" (1: XCD-balanced
 *       block -> tile tables instead of the formula map);
 *   solves / posterior / gradient: "bwd_sweep" / "fwd_sweep" (1: the backward / forward vector sweep with one right-hand side in
 *       one launch), "block_inverses" (1: the posterior substitutes with inverted diagonal blocks) ("order"), "posterior_block"
 *       (2048 / 1024: width of those blocks up to 1024 prediction points; 1024 beyond) ("order"), "posterior_halves" (1: 512-1024
 *       points as two halves on two streams) ("order"), "potri_kminor" (1: POTRI on (M,K) x (N,K) products only) ("order");
 *   diagnostics: "chain_stamps" / "leaf_stamps" (device pointers, 0 = off: in-kernel timestamps of the panel kernel's hand-offs /
 *       the leaf's phases). */
int fvgp_hip_set_option(fvgp_handle *h, const char *key, int64_t value);
/* Option "chain_verify" (0/1, default 0): every in-launch hand-off of the resident panel kernel (csrc/chain.hip: a solved block row,
 * a factored diagonal block with its tile inverses) carries a checksum of its payload, taken by the producer from what it stores and
 * compared by every consumer with what ARRIVED (LDS images of its LDS-DMA loads, registers of its buffer loads).  out2_host[0] =
 * mismatches, [1] = comparisons since the last call; synchronises.  A verifying run is slower (LDS reductions on the chain's critical
 * path) and returns the same bits.  The dpotrf it guards: gp_lin_alg.py:245. */
int fvgp_hip_chain_verify_counts(fvgp_handle *h, int64_t *out2_host);
/* out[0] = number of trailing-update launches of the last potrf, out[1] = their summed
 * duration in ms, out[2] = their summed algorithmic flops, out[3] = whole-potrf ms; of the last fused
 * evaluation (fvgp_hip_loglik): out[4] = covariance-assembly ms, out[5] = its algorithmic bytes (lower
 * 128-tiles written once), out[6] = ms of everything after the factorisation (backward solve, reductions);
 * out[7] = host milliseconds the last row-sharded evaluation (fvgp_hip_loglik_dist) took to ENQUEUE (no synchronisation inside).
 * out needs 8 doubles. */
int fvgp_hip_get_profile(fvgp_handle *h, double *out8_host);
/* the same eight, then out[8] = summed ALGORITHMIC bytes of those trailing-update launches (every C tile read and written once, the
 * panel's rows read once: what bench.py's roofline.traffic is compared with); out[9..15] reserved (0).  out needs 16 doubles. */
int fvgp_hip_get_profile_ex(fvgp_handle *h, double *out16_host);
/* The handle keeps the inverted 128 x 128 diagonal blocks of the LAST factor it produced or solved with, keyed
 * on (pointer, n, ld).  A caller that fills a factor buffer by any other means than fvgp_hip_potrf / _loglik
 * (upload of a pickled factor, bordering update, device-to-device copy: gp_kv.py:462-476,718-765) must call
 * this before the next potrs / trsm / potri / posterior on that buffer -- an allocator may hand back the
 * address of a freed factor of the same size. */
int fvgp_hip_invalidate_factor(fvgp_handle *h);

/* ---- covariance assembly -------------------------------------------------------------
 * replaces GPprior.compute_covariances -> kernel(x1,x2,hps) (gp_prior.py:217-224) and,
 * with vdiag, GPkv.addKV (gp_kv.py:639-669) fused into the same pass.
 * x1 (n1,d), x2 (n2,d) row-major device; theta host; K (rows, ldk).
 * uplo = FVGP_LOWER writes only tiles on/below the diagonal (requires x1 == x2 semantics).
 * pad != 0: K has padded_dim(n1) rows and ldk >= padded_dim(n2); padding is written
 *           (1 on the diagonal, else 0).  pad == 0: exactly n1 x n2 entries are written. */
int fvgp_hip_kmat(fvgp_handle *h, int kernel_id, const double *x1, int64_t n1, const double *x2, int64_t n2,
                  int d, const double *theta_host, int ntheta, const double *vdiag_or_null,
                  double *K, int64_t ldk, int uplo, int pad);

/* ---- dense Cholesky ------------------------------------------------------------------
 * potrf  : calculate_Chol_factor   gp_lin_alg.py:237-269   (scipy cho_factor -> dpotrf)
 * potrs  : calculate_Chol_solve    gp_lin_alg.py:289-328   (cho_solve -> dpotrs)
 * logdet : calculate_Chol_logdet   gp_lin_alg.py:331-360   (2*sum(log|diag|))
 * potri  : calculate_inv_from_chol gp_lin_alg.py:1558-1566 (KV^-1 from L), lower triangle
 * A / L: padded_dim(n) rows, lda >= padded_dim(n).  B: (padded_dim(n), ldb) row-major,
 * nrhs columns used; rows >= n of B are overwritten with zeros. */
int fvgp_hip_potrf(fvgp_handle *h, double *A, int64_t n, int64_t lda, int *info_host);
/* potrf without the host round trip (the row-sharded driver factors one diagonal block per panel and must
 * not drain the stream): info goes to `info_dev` (device int), 2*sum(log|diag|) of the first n_logdet
 * rows to `logdet_dev` (device double, may be null).  Enqueue only. */
int fvgp_hip_potrf_dev(fvgp_handle *h, double *A, int64_t n, int64_t lda, int64_t n_logdet, int *info_dev, double *logdet_dev);
int fvgp_hip_potrs(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb);
int fvgp_hip_logdet(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *out_host);
int fvgp_hip_potri(fvgp_handle *h, double *L, int64_t n, int64_t ldl, double *work, int64_t ldw);
/* forward half only: B <- L^-1 B (the new rows of an append, gp_lin_alg.py:1310-1477; the posterior covariance of kernel callables,
 * gp_posterior.py:120-136).  One column: the one-launch forward sweep; 128 .. 1024 columns against >= 2048 rows: the right-hand
 * sides are transposed into handle scratch (nrhs x padded_dim(n) doubles) and swept with the inverted 1024 x 1024 diagonal blocks. */
int fvgp_hip_trsm_lower(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb);

/* backward half only: B <- L^-T B, nrhs a multiple of 128 (GEMM path) */
int fvgp_hip_trsm_lower_t(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb);

/* ---- row-sharded (multi-GPU) building blocks: the gp2Scale partitioning pattern -----------
 * (fvgp/gp2Scale_covariance.py:381-396, gp_prior.py:319-322) applied to a DENSE factorisation:
 * 128-row blocks of K+V are dealt block-cyclically to the ranks, x is replicated.
 * panel_trsm   : P (rows, nd) <- P * L_D^-T with D the factored nd x nd diagonal block (nd % 128 == 0)
 * syrk_rowshard: C[ti][tj] -= A[ti] B[tj]^T for the 128x128 tiles with tj <= ti*scale + off, i.e. the
 *                lower-triangular part of the trailing update restricted to this rank's block rows
 *                (scale = number of ranks, off = global offset of the first local block row, may be < 0);
 *                A (M,K), B (N,K), C (M,N) row-major.  B may be used as an all-gather leaves it:
 *                b_ranks chunks of b_blocks 128-row blocks each, chunk q holding the cyclic blocks
 *                q, q + b_ranks, ...; tile column tj reads cyclic block tj + b_off
 *                (b_ranks = 1, b_off = 0: plain row order). */
int fvgp_hip_panel_trsm(fvgp_handle *h, const double *D, int64_t nd, int64_t ldd, double *P, int64_t rows, int64_t ldp);

/* ---- row-sharded (multi-GPU) evaluation: one process per GPU, RCCL over xGMI ---------------------------------
 * Stands in for the reference's distribution switch and scheduler: GP(..., gp2Scale=True, dask_client=...) (gp.py:419-439),
 * GPprior._gp2Scale_covariance / distributed_covariance (gp_prior.py:324-347, gp2Scale_covariance.py:313-431) and the
 * broadcast of x (gp_prior.py:319-322) -- with a dense factorisation behind it.
 *
 * Collectives are function pointers on device buffers, enqueued on `stream`: fvgp_hip_comm_init binds RCCL
 * (ncclAllGather / ncclAllReduce on a communicator built from the 128-byte ncclUniqueId of fvgp_hip_comm_unique_id, which
 * rank 0 hands to the others by any means -- torch.distributed's store on the Python side); fvgp_hip_comm_init_callbacks
 * binds the caller's own (the CPU tests bind gloo).  bytes = what this rank receives (bookkeeping for the timings). */
typedef struct fvgp_collectives {
    void *ctx;
    int (*all_gather)(void *ctx, const double *send, double *recv, int64_t count_per_rank, void *stream);    /* recv: nranks x count */
    int (*all_reduce_sum)(void *ctx, double *buf, int64_t count, void *stream);                             /* in place */
} fvgp_collectives;
int fvgp_hip_comm_unique_id(void *out128_host);
int fvgp_hip_comm_init(fvgp_handle *h, const void *unique_id128_host, int rank, int nranks);
int fvgp_hip_comm_init_callbacks(fvgp_handle *h, const fvgp_collectives *cb, int rank, int nranks);
int fvgp_hip_comm_destroy(fvgp_handle *h);
/* Direct collectives over peer mappings instead of a collective library (csrc/ipc.hip): the reference's workers hand covariance
 * blocks to each other directly (gp_prior.py:301-322, gp2Scale_covariance.py:419-420).  The payload moves by hipMemcpyAsync
 * between IPC mappings (copy engines, every peer at once); the only kernels are one-wave flag writers / pollers.
 *   ipc_window   : allocates this rank's window (window_bytes, two halves: a call moves at most window_bytes / 2 per piece) and
 *                  returns its 64-byte IPC handle (host); every rank hands its handle to every other by any means;
 *   comm_init_ipc: all_handles64_host = nranks x 64 bytes in rank order; shm_name = a POSIX shared-memory name ("/fvgp_...") that is
 *                  NEW for this communicator and the same on every rank (the flag words live there; unlink it once every rank has
 *                  returned).  Binds the handle's collectives like fvgp_hip_comm_init.  nranks <= 16. */
int fvgp_hip_ipc_window(fvgp_handle *h, int64_t window_bytes, void *out_handle64_host);
int fvgp_hip_comm_init_ipc(fvgp_handle *h, const void *all_handles64_host, const char *shm_name, int rank, int nranks);
/* the handle's collectives on its stream (the parts of the sharded path that are sequenced by the caller: backward solve,
 * posterior, gradient -- gp_kv.py:574-593, gp_posterior.py:139-288 on the distributed factor) */
int fvgp_hip_all_reduce(fvgp_handle *h, double *buf, int64_t count);
int fvgp_hip_all_gather(fvgp_handle *h, const double *send, double *recv, int64_t count_per_rank);
/* out[0..1] = calls, out[2..3] = bytes received, out[4..5] = milliseconds on the chain stream of the all_gather /
 * all_reduce calls since the last call of this function (option "profile" on); out needs 6 doubles */
int fvgp_hip_comm_profile(fvgp_handle *h, double *out6_host);
/* what the handle's communicator is, as the communicator itself reports it (the reference's analogue: the Dask client's own view of its
 * workers, gp_prior.py:319-322): out[0] = 0 none / 1 RCCL / 2 direct IPC / 3 caller's callbacks, out[1] = nranks and out[2] = rank as
 * bound; RCCL only: out[3] = ncclCommCount, out[4] = ncclCommUserRank, out[5] = ncclCommCuDevice, out[6] = ncclGetVersion; -1 where
 * not applicable.  out needs 8 int64. */
int fvgp_hip_comm_info(fvgp_handle *h, int64_t *out8_host);
/* 0, or 2200 once a poll of the direct (IPC) collectives has given up waiting for a peer: everything the collectives delivered since is
 * stale.  Ask after synchronising; fvgp_hip_sync and every entry that returns host values do (they fail with 2200 instead of
 * returning results computed from stale windows).  The condition is sticky: destroy the communicator and build a new one. */
int fvgp_hip_comm_check(fvgp_handle *h);

/* One GP sharded over the ranks.  Every buffer is the caller's (sizes in doubles from fvgp_hip_dist_workspace):
 *   x_all (n, d) and vdiag (n) replicated; zt (128, np): (y-m)^T in the first ncol rows, zeros below;
 *   A ((nb_max + 1) * 128, np): this rank's block rows (local block l = global block l * nranks + rank) and the block of
 *   right-hand-side rows; T[2], recv[2], Dfac, gather: panel buffers (nranks > 1 or force_general); info_dev (npan ints)
 *   and logdet_dev (npan doubles): per-panel results, on the device.
 * np = padded_dim(n), nb_max = ceil(np / 128 / nranks), npan = ceil(np / panel). */
typedef struct fvgp_dist_desc {
    int64_t n; int d; int ncol; int64_t panel; int rank, nranks; int kernel_id;
    const double *x_all; const double *vdiag; const double *zt;
    double *A; double *T[2]; double *recv[2]; double *Dfac; double *gather;
    int *info_dev; double *logdet_dev;
    int keep_factor;      /* 0: likelihood only -- the factored panels are not copied back into A (no solve can follow) */
    int force_general;    /* 1: a single rank still takes the panel-buffer / collective path (exercises RCCL on one GPU) */
    int preassembled;     /* 1: the caller has filled A's block rows with its rows of K (host kernel callables, matrix-valued noise
                           *    already added; zero padding rows): the driver adds vdiag / the identity padding and (y-m)^T only */
} fvgp_dist_desc;
/* out[0] A, [1] each T, [2] each recv, [3] Dfac, [4] gather (doubles); [5] = npan */
int fvgp_hip_dist_workspace(const fvgp_dist_desc *d, int64_t *out6);
/* GPMarginalLikelihood.log_likelihood(theta) (gp_marginal_likelihood.py:137-179) on the sharded matrix: assembly of the
 * rank's rows, blocked right-looking Cholesky with one panel of look-ahead (per panel: all-gather of the diagonal block from
 * its owners, factorisation of the tall panel, all-gather of the panel factor, trailing update of the rank's block rows),
 * the forward solve riding along.  out_host = {log-likelihood, log|KV|, (y-m)^T KV^-1 (y-m) / ncol}, replicated;
 * *info_host = dpotrf's info (global index of the first non-positive pivot) or 0.  One host synchronisation. */
int fvgp_hip_loglik_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta_host, int ntheta, double *out_host, int *info_host);
/* After fvgp_hip_loglik_dist with keep_factor = 1 -- the rank's block rows of L in A, the factored diagonal blocks replicated in
 * Dfac -- the rest of what the reference's distributed mode answers through the same object (gp_kv.py:574-593,
 * gp_posterior.py:139-288, tests/test_fvgp.py:3112-3149), each ONE call per rank (collectives issued from the library):
 *   solve_dist     : KVinvY = L^-T z, replicated: alpha_out (np x 128 row-major device; columns >= ncol are zero);
 *   posterior_dist : k^T KVinvY -> mean_out (pp x 128) and kk - k^T KV^-1 k -> S_out (pp x pp, may be null), replicated,
 *                    pp = padded_dim(npred).  Kernel callables: k_pre = the rank's rows of k(x_data, x_pred)
 *                    (nb_max*128 x pp, block l = global block l*nranks + rank, zero padded) and kk_pre = k(x_pred, x_pred)
 *                    (pp x pp, read on rank 0 only) assembled by the caller; else null and xpred (npred x d) is given;
 *   grad_dist      : 1/2 (tr(KV^-1 dK_i) - b^T dK_i b), b = alpha[:, component] (gp_marginal_likelihood.py:262-300) for the
 *                    kernel-owned hyperparameters -> grad_host (ntheta), replicated; diag_out (np device doubles or null)
 *                    receives diag(KV^-1).  The rank's rows of inv(L) (N^2 / nranks doubles) live in the scratch; the Gram
 *                    matrix is walked in column slabs of `slab` columns (a multiple of 128).
 * ws: caller-owned device scratch of fvgp_hip_dist_scratch(d, what, npred, slab) doubles, what = 0 solve, 1 posterior, 2 gradient. */
int64_t fvgp_hip_dist_scratch(const fvgp_dist_desc *d, int what, int64_t npred, int64_t slab);
int fvgp_hip_solve_dist(fvgp_handle *h, const fvgp_dist_desc *d, double *alpha_out, double *ws);
int fvgp_hip_posterior_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta_host, int ntheta, const double *xpred, int64_t npred,
                            const double *k_pre, const double *kk_pre, const double *alpha, double *mean_out, double *S_out, double *ws);
int fvgp_hip_grad_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta_host, int ntheta, const double *alpha, int component,
                       int64_t slab, double *grad_host, double *diag_out, double *ws);
/* one tall panel T (rows x w, row-major, ldt): the w x w diagonal block on top (lower triangle; the first
 * n_valid rows are data, the rest identity padding), this rank's rows of the panel below it.  Factors the top
 * block and solves the rows below against it, 128 columns at a time (leaf, TRSM by the inverted diagonal tile,
 * in-panel update), exactly as the single-GPU driver treats a panel.  Enqueue only: info -> info_dev,
 * 2*sum(log diag) of the first n_valid rows -> logdet_dev (may be null). */
int fvgp_hip_panel_potrf_dev(fvgp_handle *h, double *T, int64_t w, int64_t rows, int64_t ldt, int64_t n_valid,
                             int *info_dev, double *logdet_dev);
int fvgp_hip_syrk_rowshard(fvgp_handle *h, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                            const double *B, int64_t ldb, double *C, int64_t ldc, int scale, int off,
                            int b_ranks, int b_blocks, int b_off);

/* ---- fused evaluations ----------------------------------------------------------------
 * loglik: GPMarginalLikelihood.log_likelihood(theta)  gp_marginal_likelihood.py:137-179
 *         = kernel -> addKV -> potrf -> potrs -> logdet -> scalar, nothing leaves HBM.
 *   ymean  (n, ncol) row-major = y - m   (default mean: gp_prior.py:449-458, done by caller)
 *   KV     scratch of padded_dim(n) rows at leading dimension ld >= padded_dim(n); holds the factor L of the n x n matrix (identity
 *          on the padding) on return.  NOTHING below row padded_dim(n) is read or written, whatever ld is (a pitched buffer, a row
 *          slice of a larger arena): where n leaves fewer than ncol padding rows the forward solve is a sweep of its own.
 *          fvgp_hip_loglik_rows takes the number of rows the caller really owns: with kv_rows and ld >= fvgp_hip_loglik_dim(n, ncol)
 *          the appended (y-m)^T rows go into the extra block row and the forward solve is fused for every n (rows
 *          padded_dim(n) .. kv_rows - 1 are identity padding again on return)
 *   alpha  (padded_dim(n), ncol) receives KVinvY
 *   out_host[0] = log marginal likelihood, [1] = log|KV|, [2] = sum((y-m)*KVinvY)/ncol
 *   vdiag is REQUIRED here (n positive noise variances, gp_likelihood.py:89-110): returns -8 when NULL;
 *   every entry must be positive (the facade takes the unfused kmat/potrf/potrs route otherwise). */
int fvgp_hip_loglik(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                    const double *theta_host, int ntheta, const double *vdiag,
                    const double *ymean, int ncol, double *KV, int64_t ld,
                    double *alpha, double *out_host, int *info_host);
int fvgp_hip_loglik_rows(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                         const double *theta_host, int ntheta, const double *vdiag,
                         const double *ymean, int ncol, double *KV, int64_t kv_rows, int64_t ld,
                         double *alpha, double *out_host, int *info_host);

/* loglik_grad: GPMarginalLikelihood.neg_log_likelihood_gradient  gp_marginal_likelihood.py:224-309
 *   g_i = 1/2 sum_jk (KVinv_jk - b_j b_k) dK_jk/dtheta_i,  b = KVinvY[:,component];
 *   dK/dtheta re-evaluated on the fly (gp_prior.py:421-436, gp_bo.py:167-201).
 *   KV on entry: the factor L from fvgp_hip_loglik / potrf (destroyed: holds KV^-1 lower on return)
 *   work: second padded_dim(n) x ld scratch.  grad_host: ntheta doubles. */
int fvgp_hip_loglik_grad(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                         const double *theta_host, int ntheta, const double *alpha, int ncol, int component,
                         double *KV, int64_t ld, double *work, int64_t ldw, double *grad_host);

/* the trace part of the gradient on its own: grad_host[i] = 1/2 sum_jk (W_jk - b_j b_k) dK_jk/dtheta_i over the
 * n x n symmetric W (lower triangle read; b with stride ldb, or NULL for the pure trace 1/2 tr(W dK_i)).  The row-sharded
 * gradient calls it on each rank's partial Gram matrix inv(L)_p^T inv(L)_p (gp_marginal_likelihood.py:262-300).
 * partial: device scratch of T(T+1)/2 * ntheta doubles, T = ceil(n/128).  Synchronises. */
int fvgp_hip_grad_trace(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                        const double *theta_host, int ntheta, const double *W, int64_t ldw,
                        const double *b_or_null, int64_t ldb, double *partial, double *grad_host);
/* the same pass over a SLAB of columns [col0, col0 + ncols) of the symmetric matrix (col0 % 128 == 0): W (n, ldw) holds those
 * columns only, entries with row >= column are read.  The row-sharded gradient walks its partial Gram matrix of inv(L) slab by
 * slab, so that no rank ever holds an N x N buffer; the slabs' results add up to fvgp_hip_grad_trace's.
 * partial: ceil(n/128) * ceil(ncols/128) * ntheta doubles (device scratch). */
int fvgp_hip_grad_trace_cols(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                             const double *theta_host, int ntheta, const double *W, int64_t ldw, int64_t col0, int64_t ncols,
                             const double *b, int64_t ldb, double *partial, double *grad_host);

/* posterior: GPposterior.posterior_mean / posterior_covariance  gp_posterior.py:139-182,229-288
 *   L: factor (padded), alpha: KVinvY (padded_dim(n), ncol)
 *   kx: scratch of padded_dim(n) x ldk doubles with ldk >= padded_dim(P).  On return (P >= 2, var_out or S_out given) its first
 *       padded_dim(P) x padded_dim(n) doubles hold V^T = (L^-1 k)^T, row-major with leading dimension padded_dim(n): a caller that
 *       walks many prediction points in chunks builds the off-diagonal blocks S_ij = k(x_i, x_j) - V_i^T V_j from the chunks'
 *       scratches (fvgp_amd/gp.py _posterior_chunked; gp_posterior.py:120-136).  One point: unspecified.  The first call after a new factor
 *       also inverts its diagonal blocks into the handle (2048 x 2048 up to 1024 points, option "posterior_block"; 1024 x 1024
 *       beyond; fvgp_hip_workspace_bytes counts them).  The block width depends on P and the option only: the same call returns
 *       the same bits whether it is the first on a factor or a later one
 *   mean_out (P, ncol) device  = k^T alpha          (prior mean added by the caller)
 *   S_out (padded_dim(P), lds) device or NULL  = kk - k^T KV^-1 k  (full, symmetric)
 *   var_out (P) device  = diag of the above (unclipped; clipping is gp_posterior.py:248-259, caller side) */
/* Enqueue what the first fvgp_hip_posterior on a new factor would otherwise do in front of its sweep (the inverted diagonal blocks,
 * 2.6 ms at N = 20k) -- GPkv._refresh computes what the posterior queries need when the state changes, not at the first query
 * (gp_kv.py:404-428).  Asynchronous; a no-op where the sweep does not use inverted blocks. */
int fvgp_hip_posterior_prepare(fvgp_handle *h, const double *L, int64_t n, int64_t ldl);
int fvgp_hip_posterior(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                       const double *theta_host, int ntheta, const double *L, int64_t ldl,
                       const double *alpha, int ncol, const double *xpred, int64_t P,
                       double *kx, int64_t ldk, double *mean_out, double *var_out, double *S_out, int64_t lds);

/* ---- building blocks exported for the parity tests --------------------------------------
 * C (M,N) = alpha * opA * opB + beta * C on fp64 MFMA.  M, N multiples of 128, K of 16.
 *   a_kmajor == 0: A stored (M,K) row-major;  != 0: A stored (K,M) row-major (A^T product)
 *   b_nmajor == 0: B stored (N,K) row-major (C = A B^T); != 0: B stored (K,N) row-major
 *   lower != 0: only 128x128 tiles with row-tile >= col-tile are computed (SYRK-style)
 * With fewer than 256 output tiles and K >= 1024 the K range is split over workgroups and the partial tiles added in a fixed
 * order (handle scratch, at most 64 MB): same result on every run, another rounding than the unsplit product.  The split needs
 * C 16-byte aligned with an even ldc; a C that is not takes the unsplit product (state it if rounding must not depend on where a
 * buffer lies: pass aligned buffers). */
int fvgp_hip_gemm(fvgp_handle *h, int a_kmajor, int b_nmajor, int lower, int64_t M, int64_t N, int64_t K,
                  double alpha, const double *A, int64_t lda, const double *B, int64_t ldb,
                  double beta, double *C, int64_t ldc);
/* one 64-lane wave: D = A(16x4) * B(4x16) with the lane maps the kernels assume */
int fvgp_hip_mfma_selftest(fvgp_handle *h, const double *A16x4, const double *B4x16, double *D16x16);
/* host-only replay of the GEMM kernel's blockIdx -> (tile row, tile col) map, XCD remap included
 * (no GPU needed); returns the grid size, fills min(grid, cap) entries; out-of-range tiles are
 * the ones the kernel exits on. */
int64_t fvgp_hip_debug_tile_map(int tiles_m, int tiles_n, int lower, int scale, int off, int *out_ti, int *out_tj, int64_t cap);
/* host-only: the XCD-balanced block -> tile table the plain launches of the 128-tile kernels use instead of the formula map (block b runs
 * on XCD b % 8; each XCD gets a contiguous, equally long run of the REAL tiles in super-tile order).  Entries are
 * (tile row << 16) | tile col, or -1; returns the grid size, fills min(grid, cap) entries. */
int64_t fvgp_hip_debug_tile_table(int tiles_m, int tiles_n, int lower, int scale, int off, int *out, int64_t cap);
/* host-only: the task a start-order ticket gets in the resident panel kernel (csrc/chain.hip) for a panel of n block columns whose first
 * n2 >= n block rows have a workgroup per 128 x 128 block: out3 = {kind, block row, block column}, kind 0 = diagonal block (updates it,
 * then its leaf), 1 = block below the diagonal (products, then the solve behind the leaf), 2 = a whole block row (column = -1).
 * Per block column the diagonal block and the two blocks under it are CRITICAL (the chain of leaves runs through them), the rest BULK;
 * the critical tasks of column k + ahead are dealt right before the bulk tasks of column k ("chain_ahead").  tests/test_host_logic.py
 * replays the order: a bulk task only ever waits for LOWER tickets (ahead = 0: every task does), and the order makes progress with as
 * few as 16 slots. */
int fvgp_hip_debug_chain_ticket(int n, int n2, int ahead, int ticket, int *out3);
/* diagnostic: blocks x 256 threads each issue iters x 16 register-only fp64 MFMAs (2048 flop each per wave);
 * out needs blocks*256 doubles.  Gives the sustained fp64 MFMA ceiling of the device. */
int fvgp_hip_mfma_peak(fvgp_handle *h, double *out, int blocks, int iters);
/* A[i][j] += alpha * B[i][j] for j <= i < n: GPkv.addKV with a matrix-valued (2-d) noise model, KV = K + V
 * (gp_kv.py:654-657); only the lower triangle, like every other symmetric buffer of this ABI */
int fvgp_hip_add_lower(fvgp_handle *h, double *A, int64_t n, int64_t lda, const double *B, int64_t ldb, double alpha);
/* out = sum_{i,j<n} (W[i][j] - b_i b_j) D[i][j]  =  tr(W D) - b^T D b for symmetric W: the kernel term of the gradient,
 * dL/dtheta_i = -1/2 (b^T dK_i b - tr(KV^-1 dK_i)) (gp_marginal_likelihood.py:301-306), for a derivative matrix that exists as
 * numbers -- kernel callables (user gradient or the central differences of gp_prior.py:438-447) and matrix-valued noise
 * derivatives (gp_marginal_likelihood.py:262-267).  W, D full n x n device arrays (W symmetric: both triangles valid), b a
 * device vector with stride ldb or NULL.  Fixed-order reduction. */
int fvgp_hip_trace_dot(fvgp_handle *h, const double *W, int64_t ldw, const double *D, int64_t ldd, const double *b, int64_t ldb,
                       int64_t n, double *out_host);
/* out_host = sum_{i<n, k<c} a[i][k] b[i][k]: the data-fit term sum((y-m) o KVinvY) (gp_marginal_likelihood.py:175) */
int fvgp_hip_dot(fvgp_handle *h, const double *a, int64_t lda, const double *b, int64_t ldb, int64_t n, int c, double *out_host);
/* out[p] = sum_i A[i][p] B[i][p], p < cols (device): einsum('ij,jk,ki->i', k^T, KVinv, k) of the CholInv variance path
 * (gp_posterior.py:238-244) after KVinv k came from fvgp_hip_gemm */
int fvgp_hip_coldot(fvgp_handle *h, const double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double *out);
/* out[p] = sum_i V[i][p]^2, p < ncols: the column sums of squares of a rank's rows of inv(L) are its share of diag(KV^-1)
 * (gradients of noise-function hyperparameters, gp_marginal_likelihood.py:262-267, in the row-sharded mode) */
int fvgp_hip_colsumsq(fvgp_handle *h, const double *V, int64_t rows, int64_t ldv, int64_t ncols, double *out);
/* A[i][j] += alpha * B[i][j], rows x cols: a rank's rows of a matrix-valued noise model added to its rows of K (gp_kv.py:654-657) */
int fvgp_hip_add_matrix(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double alpha);
/* mirror the lower triangle into the upper (for exporting K / KV^-1 to numpy) */
int fvgp_hip_symmetrize(fvgp_handle *h, double *A, int64_t n, int64_t lda);

#ifdef __cplusplus
}
#endif
#endif
