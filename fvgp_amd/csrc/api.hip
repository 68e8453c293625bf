// Host side of the C ABI (include/fvgp_hip.h): argument checks, the blocked drivers that
// sequence the kernels on the handle's stream, and the fused log-likelihood / gradient /
// posterior evaluations.  No torch types; plain pointers and sizes only.
#include "common.h"
#include <math.h>
#include <string.h>

// ---------------------------------------------------------------------------------------
static thread_local std::string g_err;
void fvgp_set_error(const std::string &s) { g_err = s; }
int fvgp_hip_fail(hipError_t e, const char *what, int line) {
    g_err = std::string("HIP error '") + hipGetErrorString(e) + "' at api line " + std::to_string(line) + ": " + what;
    return 1000 + (int)e;
}

extern "C" {

int fvgp_hip_version(void) { return 100; }
const char *fvgp_hip_last_error_string(void) { return g_err.c_str(); }
int64_t fvgp_hip_workspace_bytes(int64_t n, int64_t npred) {
    // what a handle allocates on the device for problems of n points (and npred prediction points):
    // inverted 128 x 128 diagonal blocks + per-leaf log-det partials + the solve / posterior scratch + reductions
    if (n <= 0 || npred < 0) return -1;
    const int64_t np = pad128(n), nblk = np / TILE, pp = pad128(npred);
    int64_t vec = np * 16 + pp * 16 > np * 8 ? np * 16 + pp * 16 : np * 8;      // posterior mean widening vs vector sweeps
    int64_t winv = 0;
    if (npred > 0) {
        // posterior: inverted diagonal blocks (np x WB; 2048 wide up to 1024 points, 1024 beyond), their doubling
        // scratch (np x WB/4), one block of the transposed right-hand sides + its split-K partials, split-K partials of S -= V^T V
        const int64_t WB = pp <= 1024 ? 2048 : 1024;
        winv = np * WB;
        const int64_t tiles = (pp / TILE) * (WB / TILE), want = tiles >= 512 ? 1 : 512 / tiles;
        const int64_t stiles = (pp / TILE) * (pp / TILE + 1) / 2, swant = stiles >= 512 ? 1 : 64 / ((stiles + 7) / 8);      // S -= V^T V: lower tiles
        const int64_t cand[3] = {np * (WB / 4) + 1024, (1 + want) * pp * WB + 64, swant * pp * pp + 64};
        for (int64_t c : cand) if (c > vec) vec = c;
    }
    // + the per-CU yield counters, the backward sweep's granules (16 bytes per row) and ticket, the resident panel kernel's flag words
    const int64_t round3 = (int64_t)CU_YIELD_KEYS * CU_YIELD_STRIDE * (int64_t)sizeof(int) + np * 16 + 2 * (int64_t)sizeof(int) + 80 * 16 * (int64_t)sizeof(unsigned long long);
    return (nblk * LEAF_DOUBLES + nblk * TILE + (npred > 0 ? vec : np * 8) + winv + RED_SLOTS) * (int64_t)sizeof(double) + (int64_t)sizeof(int) + round3;
}

int64_t fvgp_hip_padded_dim(int64_t n) { return pad128(n); }
int64_t fvgp_hip_loglik_dim(int64_t n, int ncol) {
    if (n <= 0 || ncol < 1) return -1;
    return (pad128(n) - n) >= ncol ? pad128(n) : pad128(n + ncol);
}

int fvgp_hip_create(fvgp_handle **out, int device, void *stream) {
    if (!out) return -1;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { fvgp_set_error("no such HIP device"); return -2; }
    HIPCHK(hipSetDevice(device));
    fvgp_handle *h = new fvgp_handle();
    h->device = device;
    h->stream = (hipStream_t)stream;
    HIPCHK(hipMalloc((void **)&h->red, RED_SLOTS * sizeof(double)));
    HIPCHK(hipMalloc((void **)&h->dinfo, 64));
    HIPCHK(hipMalloc((void **)&h->cu_yield, (size_t)CU_YIELD_KEYS * CU_YIELD_STRIDE * sizeof(int)));
    HIPCHK(hipMemset(h->cu_yield, 0, (size_t)CU_YIELD_KEYS * CU_YIELD_STRIDE * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&h->hpin, RED_SLOTS * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, device));
    *out = h;
    return 0;
}

int fvgp_hip_destroy(fvgp_handle *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->side) (void)hipStreamSynchronize(h->side);      // the look-ahead / chain stream may still read the buffers freed below
    for (auto e : h->ev) (void)hipEventDestroy(e);
    if (h->ev_panel) (void)hipEventDestroy(h->ev_panel);
    for (auto e : h->ev_stage) if (e) (void)hipEventDestroy(e);
    for (auto e : h->rs_ev) (void)hipEventDestroy(e);
    if (h->ev_cols) (void)hipEventDestroy(h->ev_cols);
    gemm_release_tables(h);
    (void)fvgp_hip_comm_destroy(h);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->linv) (void)hipFree(h->linv);
    if (h->winv) (void)hipFree(h->winv);
    if (h->logdet_parts) (void)hipFree(h->logdet_parts);
    if (h->red) (void)hipFree(h->red);
    if (h->dinfo) (void)hipFree(h->dinfo);
    if (h->cu_yield) (void)hipFree(h->cu_yield);
    if (h->tr_ws) (void)hipFree(h->tr_ws);
    if (h->sweep_gran) (void)hipFree(h->sweep_gran);
    if (h->sweep_ticket) (void)hipFree(h->sweep_ticket);
    if (h->chain_flags) (void)hipFree(h->chain_flags);
    if (h->chain_vhash) (void)hipFree(h->chain_vhash);
    if (h->vec) (void)hipFree(h->vec);
    if (h->hpin) (void)hipHostFree(h->hpin);
    delete h;
    return 0;
}

int fvgp_hip_stream_create(void **out_stream, int device, int high_priority, const uint32_t *cu_mask, int mask_words) {
    if (!out_stream) return -1;
    if (cu_mask && mask_words < 1) return -5;
    HIPCHK(hipSetDevice(device));
    hipStream_t s = nullptr;
    if (cu_mask) {
        HIPCHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask));
    } else {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo));
    }
    *out_stream = s;
    return 0;
}

int fvgp_hip_stream_destroy(void *stream) {
    if (!stream) return -1;
    HIPCHK(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

int fvgp_hip_sync(fvgp_handle *h) {
    if (!h) return -1;
    HIPCHK(hipStreamSynchronize(h->stream));
    return fvgp_ipc_check(h);
}

int fvgp_hip_set_option(fvgp_handle *h, const char *key, int64_t value) {
    if (!h) return -1;
    if (!key) return -2;
    if (!strcmp(key, "outer_block")) {
        if (value < 128 || value % 128) { fvgp_set_error("outer_block must be a positive multiple of 128"); return -3; }
        h->outer_block = value;
        return 0;
    }
    if (!strcmp(key, "schedule")) {
        // THE schedule switch: one of the three factorisation schedules, each with the settings it was tuned with; every other
        // schedule key below is a measurement knob underneath one of them
        if (value == 0) { h->chain_wide = 1; h->lookahead_min = (int64_t)1 << 40; h->panel_chain = 1; h->lookahead = 1; }            // wide: 4096-wide panels, each alone on the chip
        else if (value == 1) { h->chain_wide = 0; h->lookahead_min = 4608; h->panel_chain = 1; h->lookahead = 1; }                  // lookahead: 2048/1024/512 panels, the next one under the update (the row-sharded driver's form)
        else if (value == 2) { h->chain_wide = 0; h->lookahead_min = 4608; h->panel_chain = 0; h->lookahead = 1; }                  // narrow: the same panels, three launches per 128 columns instead of the resident kernel
        else { fvgp_set_error("schedule: 0 = wide (default), 1 = lookahead, 2 = narrow"); return -3; }
        return 0;
    }
    if (!strcmp(key, "profile")) { h->profile = value ? 1 : 0; return 0; }
    if (!strcmp(key, "lookahead")) { h->lookahead = value ? 1 : 0; return 0; }
    if (!strcmp(key, "outer_block_big")) {
        if (value != 0 && (value < 128 || value % 128)) { fvgp_set_error("outer_block_big must be 0 or a multiple of 128"); return -3; }
        h->outer_block_big = value; return 0;
    }
    if (!strcmp(key, "big_threshold")) { h->big_threshold = value; return 0; }
    if (!strcmp(key, "tile_tables")) { h->tile_tables = value ? 1 : 0; return 0; }
    if (!strcmp(key, "chain_stamps")) { h->chain_stamps = reinterpret_cast<unsigned long long *>((uintptr_t)value); h->chain_seq = 0; return 0; }
    if (!strcmp(key, "leaf_stamps")) { h->leaf_stamps = reinterpret_cast<unsigned long *>((uintptr_t)value); return 0; }
    if (!strcmp(key, "small_tile_max")) { h->small_tile_max = value; return 0; }
    if (!strcmp(key, "small_tile_max_update")) { h->small_tile_max_update = value; return 0; }
    if (!strcmp(key, "inner_block")) {
        if (value != 0 && (value < 128 || value % 128)) { fvgp_set_error("inner_block must be 0 or a multiple of 128"); return -3; }
        h->inner_block = value; return 0;
    }
    if (!strcmp(key, "panel_recursive")) { h->panel_recursive = value ? 1 : 0; return 0; }
    if (!strcmp(key, "leaf_tiles")) { h->leaf_tiles = value ? 1 : 0; return 0; }
    if (!strcmp(key, "leaf_tiles_rows")) { h->leaf_tiles_rows = value; return 0; }
    if (!strcmp(key, "k128_kernels")) { h->k128_kernels = value ? 1 : 0; return 0; }
    if (!strcmp(key, "block_inverses")) { h->block_inverses = value ? 1 : 0; return 0; }
    if (!strcmp(key, "potri_kminor")) { h->potri_kminor = value ? 1 : 0; return 0; }
    if (!strcmp(key, "leaf_yield")) { h->leaf_yield = (int)value; return 0; }
    if (!strcmp(key, "chain_yield")) { h->chain_yield = (int)value; return 0; }
    if (!strcmp(key, "lookahead_min")) { h->lookahead_min = value; return 0; }
    if (!strcmp(key, "panel_chain")) { if (value < 0 || value > 3) return -3; h->panel_chain = (int)value; return 0; }
    if (!strcmp(key, "panel_chain_min")) { h->panel_chain_min = value; return 0; }
    if (!strcmp(key, "chain_wide")) { h->chain_wide = (int)value; return 0; }
    if (!strcmp(key, "chain_sleep_rows")) { h->chain_sleep_rows = (int)value; return 0; }
    if (!strcmp(key, "chain_ahead")) { if (value < 0 || value > 8) return -3; h->chain_ahead = (int)value; return 0; }
    if (!strcmp(key, "chain_single_rows")) { h->chain_single_rows = (int)value; return 0; }
    if (!strcmp(key, "wide_block") || !strcmp(key, "wide_block_big")) {
        if (value < TILE || value % TILE || value / TILE > FVGP_CHAIN_MAX_BLOCKS) { fvgp_set_error("wide_block: a multiple of 128, at most 4096"); return -2; }
        (key[10] ? h->wide_block_big : h->wide_block) = value; return 0;
    }
    if (!strcmp(key, "wide_threshold")) { h->wide_threshold = value; return 0; }
    if (!strcmp(key, "wide_inner")) { if (value % TILE) return -2; h->wide_inner = value; return 0; }
    if (!strcmp(key, "wide_inner_rows")) { h->wide_inner_rows = value; return 0; }
    if (!strcmp(key, "cols_split")) { h->cols_split = value ? 1 : 0; return 0; }
    if (!strcmp(key, "chain_verify")) { h->chain_verify = value ? 1 : 0; return 0; }
    if (!strcmp(key, "cols_split_rows")) { h->cols_split_rows = value; return 0; }
    if (!strcmp(key, "bwd_sweep")) { h->bwd_sweep = (int)value; return 0; }
    if (!strcmp(key, "fwd_sweep")) { h->fwd_sweep = (int)value; return 0; }
    if (!strcmp(key, "posterior_halves")) { h->posterior_halves = (int)value; return 0; }
    if (!strcmp(key, "posterior_block")) { if (value != 1024 && value != 2048) return -3; h->posterior_block = value; return 0; }
    if (!strcmp(key, "outer_block_small")) { if (value < 0 || value % TILE) return -3; h->outer_block_small = value; return 0; }
    if (!strcmp(key, "small_threshold")) { h->small_threshold = value; return 0; }
    fvgp_set_error(std::string("unknown option ") + key);
    return -2;
}

int fvgp_hip_invalidate_factor(fvgp_handle *h) {
    if (!h) return -1;
    h->winv_ok = false; h->linv_L = nullptr;
    return 0;
}

int fvgp_hip_chain_verify_counts(fvgp_handle *h, int64_t *out2_host) {
    if (!h) return -1;
    if (!out2_host) return -2;
    HIPCHK(hipSetDevice(h->device));
    unsigned long long w[2];
    const int rc = chain_verify_counts(h, w);
    out2_host[0] = (int64_t)w[0]; out2_host[1] = (int64_t)w[1];
    return rc;
}

int fvgp_hip_get_profile(fvgp_handle *h, double *out) {
    if (!h) return -1;
    if (!out) return -2;
    if (h->rs_used > 0) {
        // the row-sharded driver enqueues its trailing updates one ABI call at a time: their events are read here
        HIPCHK(hipStreamSynchronize(h->stream));
        h->prof_launches = (double)h->rs_flops.size(); h->prof_ms = 0; h->prof_flops = 0;
        for (size_t i = 0; i < h->rs_flops.size(); ++i) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->rs_ev[2 * i], h->rs_ev[2 * i + 1]));
            h->prof_ms += ms; h->prof_flops += h->rs_flops[i];
        }
        h->rs_used = 0; h->rs_flops.clear();
    }
    out[0] = h->prof_launches; out[1] = h->prof_ms; out[2] = h->prof_flops; out[3] = h->prof_total_ms;
    out[4] = h->prof_kmat_ms; out[5] = h->prof_kmat_bytes; out[6] = h->prof_tail_ms; out[7] = h->prof_host_enqueue_ms;
    return 0;
}

int fvgp_hip_get_profile_ex(fvgp_handle *h, double *out16) {
    int rc = fvgp_hip_get_profile(h, out16); if (rc) return rc;
    for (int i = 8; i < 16; ++i) out16[i] = 0.0;
    out16[8] = h->prof_bytes;
    return 0;
}

}  // extern "C"

// the high-priority second stream (look-ahead panel chain) and the two events that order it against the main stream
int fvgp_ensure_side(fvgp_handle *h) {
    if (h->side) return 0;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));
    HIPCHK(hipEventCreateWithFlags(&h->ev_panel, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_cols, hipEventDisableTiming));
    return 0;
}

// ---------------------------------------------------------------------------------------
static int ensure_blocks(fvgp_handle *h, int64_t nblk) {
    if ((size_t)nblk > h->linv_blocks) {
        if (h->linv) HIPCHK(hipFree(h->linv));
        h->linv = nullptr; h->linv_blocks = 0; h->winv_ok = false; h->linv_L = nullptr;
        HIPCHK(hipMalloc((void **)&h->linv, (size_t)nblk * LEAF_DOUBLES * sizeof(double)));
        h->linv_blocks = (size_t)nblk;
    }
    if ((size_t)nblk > h->logdet_cap) {
        if (h->logdet_parts) HIPCHK(hipFree(h->logdet_parts));
        h->logdet_parts = nullptr; h->logdet_cap = 0;
        HIPCHK(hipMalloc((void **)&h->logdet_parts, (size_t)nblk * TILE * sizeof(double)));
        h->logdet_cap = (size_t)nblk;
    }
    return 0;
}

int ensure_scratch(fvgp_handle *h, int64_t np) {
    size_t need = (size_t)np * 8;
    if (need > h->vec_cap) {
        if (h->vec) HIPCHK(hipFree(h->vec));
        h->vec = nullptr; h->vec_cap = 0;
        HIPCHK(hipMalloc((void **)&h->vec, need * sizeof(double)));
        h->vec_cap = need;
    }
    return 0;
}

// diagonal-block inverses for factor L: reuse those left by potrf, else recompute (batched)
int ensure_linv(fvgp_handle *h, const double *L, int64_t n, int64_t ldl) {
    const int64_t np = pad128(n), nblk = np / TILE;
    if (h->linv_L == L && h->linv_n == n && h->linv_ld == ldl && (size_t)nblk <= h->linv_blocks) return 0;
    int rc = ensure_blocks(h, nblk);
    if (rc) return rc;
    rc = launch_leaf_inverse_batched(h, L, ldl, nblk, h->linv);
    if (rc) return rc;
    h->winv_ok = false; h->linv_L = L; h->linv_n = n; h->linv_ld = ldl;
    return 0;
}

// inverses of the WB x WB diagonal blocks of L (WB = 1024: POTRI, posterior at more than 1024 points; 2048: posterior up to 1024
// points), from the 128-block inverses by doubling:
//     inv [[A, 0], [C, B]] = [[inv A, 0], [-inv(B) C inv(A), inv B]]      at block sizes 128 -> 256 -> 512 -> 1024 (-> 2048),
// every level two strided-batch GEMM launches over all full WB-blocks (T = C inv(A) into the handle scratch, then
// -inv(B) T into place) plus single launches for the pairs of a narrower last block.  O(N WB^2) flops, a few hundred
// microseconds; kept until the factor changes or the other width is asked for.
// `upto` <= WB: the doubling stops at upto x upto blocks (they sit on the diagonal of the WB-wide layout); a later call with a
// larger `upto` only adds the missing levels.
static int ensure_winv(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, int64_t WB = 1024, int64_t upto = 0) {
    if (upto <= 0 || upto > WB) upto = WB;
    int rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
    if (h->winv_ok && h->winv_w == WB && h->winv_level >= upto) return 0;
    const bool extend = h->winv_ok && h->winv_w == WB;          // the lower levels are there
    h->winv_ok = false;
    const int64_t np = pad128(n), nblk = np / TILE;
    const size_t need = (size_t)np * WB;
    if (need > h->winv_cap) {
        if (h->winv) HIPCHK(hipFree(h->winv));
        h->winv = nullptr; h->winv_cap = 0;
        HIPCHK(hipMalloc((void **)&h->winv, need * sizeof(double)));
        h->winv_cap = need;
    }
    double *W = h->winv;
    if (!extend) { rc = launch_winv_seed(h, h->linv, nblk, W, WB); if (rc) return rc; }
    rc = ensure_scratch(h, np * (WB / 32) + 128); if (rc) return rc;   // T: at most np/2 x WB/2 doubles
    double *T = h->vec;
    const int64_t nfull = np / WB, t0 = nfull * WB, wt = np - t0;
    for (int64_t hs = extend ? h->winv_level : TILE; hs < upto; hs *= 2) {
        const int64_t ny = WB / (2 * hs);
        if (nfull > 0) {
            GemmDesc a{};   // T[y, z] = C inv(A)
            a.a_kmajor = 0; a.b_nmajor = 1; a.lower = 0; a.M = hs; a.N = hs; a.K = hs; a.alpha = 1.0; a.beta = 0.0;
            a.A = L + hs * ldl; a.lda = ldl; a.B = W; a.ldb = WB; a.C = T; a.ldc = hs;
            a.batch_y = (int)ny; a.batch_z = (int)nfull;
            a.a_by = 2 * hs * ldl + 2 * hs; a.a_bz = WB * ldl + WB;
            a.b_by = 2 * hs * WB + 2 * hs; a.b_bz = WB * WB;
            a.c_by = hs * hs; a.c_bz = ny * hs * hs;
            rc = launch_gemm(h, a); if (rc) return rc;
            GemmDesc b{};   // W21[y, z] = -inv(B) T
            b.a_kmajor = 0; b.b_nmajor = 1; b.lower = 0; b.M = hs; b.N = hs; b.K = hs; b.alpha = -1.0; b.beta = 0.0;
            b.A = W + hs * WB + hs; b.lda = WB; b.B = T; b.ldb = hs; b.C = W + hs * WB; b.ldc = WB;
            b.batch_y = (int)ny; b.batch_z = (int)nfull;
            b.a_by = 2 * hs * WB + 2 * hs; b.a_bz = WB * WB;
            b.b_by = hs * hs; b.b_bz = ny * hs * hs;
            b.c_by = 2 * hs * WB + 2 * hs; b.c_bz = WB * WB;
            rc = launch_gemm(h, b); if (rc) return rc;
        }
        double *Tt = T + nfull * ny * hs * hs;                           // the last, narrower block: its pairs one by one
        for (int64_t s = 0; s + hs < wt; s += 2 * hs) {
            const int64_t wb = (wt - s - hs < hs) ? wt - s - hs : hs;
            GemmDesc a{};
            a.a_kmajor = 0; a.b_nmajor = 1; a.lower = 0; a.M = wb; a.N = hs; a.K = hs; a.alpha = 1.0; a.beta = 0.0;
            a.A = L + (t0 + s + hs) * ldl + t0 + s; a.lda = ldl; a.B = W + (t0 + s) * WB + s; a.ldb = WB; a.C = Tt; a.ldc = hs;
            rc = launch_gemm(h, a); if (rc) return rc;
            GemmDesc b{};
            b.a_kmajor = 0; b.b_nmajor = 1; b.lower = 0; b.M = wb; b.N = hs; b.K = wb; b.alpha = -1.0; b.beta = 0.0;
            b.A = W + (t0 + s + hs) * WB + s + hs; b.lda = WB; b.B = Tt; b.ldb = hs; b.C = W + (t0 + s + hs) * WB + s; b.ldc = WB;
            rc = launch_gemm(h, b); if (rc) return rc;
        }
    }
    h->winv_ok = true; h->winv_w = WB; h->winv_level = upto;
    return 0;
}

static int check_square(const void *A, int64_t n, int64_t ld, int argA, int argn, int argld) {
    if (!A) return -argA;
    if (n <= 0) return -argn;
    if (ld < pad128(n) || (ld & 1)) { fvgp_set_error("leading dimension must be even and >= padded_dim(n)"); return -argld; }
    if ((uintptr_t)A & 15) { fvgp_set_error("matrix base must be 16-byte aligned"); return -argA; }
    return 0;
}

// ---------------------------------------------------------------------------------------
// one (sub-)panel of the blocked Cholesky, 128 columns at a time: leaf (potf2 + trtri in LDS) -> panel TRSM as a
// GEMM with inv(L_kk) -> update of the remaining columns of this (sub-)panel (K = 128)
static int panel_factor(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda, int64_t J0, int64_t Jend) {
    int rc;
    for (int64_t k0 = J0; k0 < Jend; k0 += TILE) {
        const int64_t kb = k0 / TILE;
        const int64_t nv = n - k0;
        const int64_t r0 = k0 + TILE, R = np - r0;
        // few rows below = the chain is what the factorisation waits for: the leaf then skips the triangular inverse of its
        // block (19 of 103 thousand cycles) and the TRSM substitutes with the 16 x 16 tile inverses; with many rows the
        // product with the full inverse is the cheaper TRSM and the leaf is hidden under the trailing update anyway
        const int tiles = (h->leaf_tiles && R <= h->leaf_tiles_rows) ? 1 : 0;
        rc = launch_leaf(h, A + k0 * lda + k0, lda, h->linv + kb * LEAF_DOUBLES, h->logdet_parts + kb * TILE, (int)k0, 1,
                         nv >= TILE ? TILE : (nv > 0 ? (int)nv : 0), tiles);
        if (rc) return rc;
        if (R <= 0) continue;
        // panel TRSM in place: A[r0:, k0:k0+128] <- A[r0:, k0:k0+128] * inv(L_kk)^T
        if (tiles) {        // by substitution with the inverses of the diagonal block's 16 x 16 tiles (all the leaf left)
            rc = launch_trsm_tiles(h, A + r0 * lda + k0, lda, R, A + k0 * lda + k0, lda, h->linv + kb * LEAF_DOUBLES);
            if (rc) return rc;
        } else {
            GemmDesc t{};
            t.a_kmajor = 0; t.b_nmajor = 0; t.lower = 0; t.M = R; t.N = TILE; t.K = TILE;
            t.alpha = 1.0; t.beta = 0.0;
            t.A = A + r0 * lda + k0; t.lda = lda;
            t.B = h->linv + kb * LEAF_DOUBLES; t.ldb = TILE;
            t.C = A + r0 * lda + k0; t.ldc = lda;
            rc = launch_gemm(h, t);
            if (rc) return rc;
        }
        // update of the rest of the outer panel: A[r0:, r0:Jend] -= P P[0:Jend-r0]^T (lower tiles)
        const int64_t W = Jend - r0;
        if (W > 0) {
            GemmDesc u{};
            u.a_kmajor = 0; u.b_nmajor = 0; u.lower = 1; u.M = R; u.N = W; u.K = TILE;
            u.alpha = -1.0; u.beta = 1.0;
            u.A = A + r0 * lda + k0; u.lda = lda;
            u.B = A + r0 * lda + k0; u.ldb = lda;
            u.C = A + r0 * lda + r0; u.ldc = lda;
            rc = launch_gemm(h, u);
            if (rc) return rc;
        }
    }
    return 0;
}

// trailing update with the factored panel [J0, Jend): block columns [c0, c1) of the trailing matrix
// (rows c0..np), lower tiles only:  A[c0:, c0:c1] -= L[c0:, J0:Jend] L[c0:c1, J0:Jend]^T
static int trailing_update(fvgp_handle *h, double *A, int64_t np, int64_t lda, int64_t J0, int64_t Jend, int64_t c0, int64_t c1, int role = 1,
                           bool *big_kernel = nullptr) {
    if (big_kernel) *big_kernel = false;
    if (c1 <= c0 || np <= c0) return 0;
    GemmDesc s{};
    s.a_kmajor = 0; s.b_nmajor = 0; s.lower = 1; s.M = np - c0; s.N = c1 - c0; s.K = Jend - J0;
    s.alpha = -1.0; s.beta = 1.0; s.role = role;
    s.A = A + c0 * lda + J0; s.lda = lda;
    s.B = A + c0 * lda + J0; s.ldb = lda;
    s.C = A + c0 * lda + c0; s.ldc = lda;
    // the last update behind a wide panel may have a handful of tiles and K = 4096 (N = 4096 with its extra block row: ONE tile, 0.2 ms
    // on four compute units): split K so that the launch fills the chip once, partial tiles summed in a fixed order
    const int64_t tm = s.M / TILE, tn = s.N / TILE, tiles = tn * (tn + 1) / 2 + (tm - tn) * tn;
    if (role == 1 && h->chain_alone && h->chain_wide && tiles > 0 && tiles <= 64 && s.K >= 1024) {
        int64_t split = 256 / tiles;
        if (split > s.K / 512) split = s.K / 512;
        if (split > 1) {
            int rc = ensure_scratch(h, (split * s.M * s.N + 7) / 8); if (rc) return rc;
            s.split = (int)split; s.split_ws = h->vec;
            if (big_kernel) *big_kernel = true;
            return launch_gemm(h, s);
        }
    }
    if (big_kernel) *big_kernel = !gemm_takes_small_tiles(h, s);
    return launch_gemm(h, s);
}

// a panel wider than `inner_block` is factored in sub-panels of that width: 128-column steps inside a sub-panel,
// then one update of the remaining columns of the panel with K = inner_block -- a third block size between the
// leaf (128) and the trailing update (panel width), so that wide panels do not pay for their width in K = 128 work
// One 128-column step of the chain without the in-panel update: leaf, then the TRSM of every row below.
static int panel_step(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda, int64_t k0) {
    return panel_factor(h, A, n, np, lda, k0, k0 + TILE);
}

// Recursive panel: left half, ONE update of the right half's columns with K = width of the left half, right half.  Against
// the right-looking loop of panel_factor (after every 128 columns an update of ALL remaining columns of the panel with
// K = 128) the same flops make 2/3 of the read-modify-write passes over the panel's columns at the 512 level and run at
// K = 256 / 512 / 1024 where they can; the update right before a leaf only touches the columns that leaf needs.
static int panel_factor_recursive(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda, int64_t J0, int64_t Jend) {
    const int64_t blocks = (Jend - J0) / TILE;
    if (blocks <= 1) return panel_step(h, A, n, np, lda, J0);
    const int64_t mid = J0 + (blocks / 2) * TILE;
    int rc = panel_factor_recursive(h, A, n, np, lda, J0, mid); if (rc) return rc;
    rc = trailing_update(h, A, np, lda, J0, mid, mid, Jend, 0); if (rc) return rc;
    return panel_factor_recursive(h, A, n, np, lda, mid, Jend);
}

static int panel_factor_nested(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda, int64_t J0, int64_t Jend) {
    if (h->panel_recursive) return panel_factor_recursive(h, A, n, np, lda, J0, Jend);
    const int64_t inner = h->inner_block;
    if (inner <= 0 || Jend - J0 <= inner) return panel_factor(h, A, n, np, lda, J0, Jend);
    for (int64_t s0 = J0; s0 < Jend; s0 += inner) {
        const int64_t s1 = (s0 + inner < Jend) ? s0 + inner : Jend;
        int rc = panel_factor(h, A, n, np, lda, s0, s1); if (rc) return rc;
        if (s1 < Jend) { rc = trailing_update(h, A, np, lda, s0, s1, s1, Jend, 0); if (rc) return rc; }
    }
    return 0;
}

// one panel, every row from its first column down: the resident panel kernel (chain.hip) while enough rows remain for a
// trailing update to run beside it, else the three launches per 128 columns
static int panel_factor_any(fvgp_handle *h, double *A, int64_t n, int64_t np, int64_t lda, int64_t J0, int64_t Jend) {
    // (the resident kernel has flag words for 32 block columns: a wider panel -- `outer_block` above 4096 -- takes the nested chain)
    if (h->panel_chain && (np - J0 >= h->panel_chain_min || (h->chain_alone && h->chain_wide)) && (Jend - J0) / TILE <= FVGP_CHAIN_MAX_BLOCKS) {
        // a tall panel (alone on the chip): sub-panels of `wide_inner` columns by the resident kernel, the rest of the panel's columns
        // brought up to date by the trailing update's kernel with K = wide_inner in between -- three quarters of the panel's flops
        // move from the resident kernel's products (0.7 of the MFMA rate) to that kernel (0.92)
        const int64_t inner = h->wide_inner;
        if (h->chain_alone && inner > 0 && Jend - J0 > inner && np - J0 >= h->wide_inner_rows) {
            for (int64_t s0 = J0; s0 < Jend; s0 += inner) {
                const int64_t s1 = (s0 + inner < Jend) ? s0 + inner : Jend;
                int rc = launch_panel_chain(h, A, n, np, lda, s0, s1); if (rc) return rc;
                if (s1 < Jend) { rc = trailing_update(h, A, np, lda, s0, s1, s1, Jend, 0); if (rc) return rc; }
            }
            return 0;
        }
        return launch_panel_chain(h, A, n, np, lda, J0, Jend);
    }
    return panel_factor_nested(h, A, n, np, lda, J0, Jend);
}

// algorithmic bytes of a lower-tile update: every C tile read and written once, the panel's rows (the B operand is the top of A) read once
static double lower_bytes(int64_t M, int64_t N, int64_t K) {
    const double tm = (double)(M / TILE), tn = (double)(N / TILE);
    const double tiles = tn * (tn + 1.0) * 0.5 + (tm - tn) * tn;
    return tiles * 128.0 * 128.0 * 8.0 * 2.0 + (double)M * (double)K * 8.0;
}

static double lower_flops(int64_t M, int64_t N, int64_t K) {     // algorithmic flops of a lower-tile update
    const double tm = (double)(M / TILE), tn = (double)(N / TILE);
    const double tiles = tn * (tn + 1.0) * 0.5 + (tm - tn) * tn;
    return tiles * 128.0 * 128.0 * 2.0 * (double)K;
}

// ---------------------------------------------------------------------------------------
// blocked right-looking Cholesky, three block sizes:
//   128        : panel_factor above;
//   inner_block: a wide outer panel is factored in sub-panels of this width, each followed by one update of the rest
//                of the outer panel with K = inner_block (panel_factor_nested);
//   outer NB   : one trailing SYRK per outer panel with K = NB (outer_block, or outer_block_big while more than
//                big_threshold rows remain), which carries ~all the flops and keeps the C-tile read-modify-write
//                traffic at 8/NB bytes per flop.
// look-ahead (option "lookahead"): the trailing update of panel J is split into the block columns of
// panel J+1 (done first) and the rest; panel J+1 is then factored on a second, high-priority stream
// while the rest of the update runs on the main stream.
// np_force != 0: the padded matrix has np_force rows (fvgp_hip_loglik appends (y-m)^T in a block row of its own when n leaves no padding rows)
// skip_inverses: the caller launches the batched block inverses itself (fvgp_hip_loglik: after it has taken the appended rows out again)
static int potrf_driver(fvgp_handle *h, double *A, int64_t n, int64_t lda, int *info_host, int *info_dev = nullptr, bool enqueue_only = false,
                        int64_t np_force = 0, bool skip_inverses = false) {
    const int64_t np = np_force ? np_force : pad128(n), nblk = np / TILE;
    int rc = ensure_blocks(h, nblk);
    if (rc) return rc;
    h->winv_ok = false; h->linv_L = nullptr;
    HIPCHK(hipMemsetAsync(h->dinfo, 0, sizeof(int), h->stream));
    const int64_t NB = h->outer_block;
    size_t nev = 0;
    h->ev_flops.clear(); h->ev_bytes.clear();
    hipEvent_t e_begin = nullptr, e_end = nullptr;
    auto get_event = [&](hipEvent_t *e) -> int {
        if (nev >= h->ev.size()) { hipEvent_t x; HIPCHK(hipEventCreate(&x)); h->ev.push_back(x); }
        *e = h->ev[nev++];
        return 0;
    };
    const bool profile = h->profile && !enqueue_only;
    if (profile) { rc = get_event(&e_begin); if (rc) return rc; HIPCHK(hipEventRecord(e_begin, h->stream)); }
    auto timed_update = [&](int64_t J0, int64_t Jend, int64_t c0, int64_t c1) -> int {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) { int r = get_event(&e0); if (r) return r; HIPCHK(hipEventRecord(e0, h->stream)); }
        bool big = false;
        int r = trailing_update(h, A, np, lda, J0, Jend, c0, c1, 1, &big);
        if (r) return r;
        if (profile) {
            // only launches of the kernel the roofline names count (sub-round updates run the small-tile kernel): 0 flops = skipped
            r = get_event(&e1); if (r) return r; HIPCHK(hipEventRecord(e1, h->stream));
            h->ev_flops.push_back(big ? lower_flops(np - c0, c1 - c0, Jend - J0) : 0.0);
            h->ev_bytes.push_back(big ? lower_bytes(np - c0, c1 - c0, Jend - J0) : 0.0);
        }
        return 0;
    };

    // panel boundaries: width NB, or the wider `outer_block_big` while more than `big_threshold` rows remain
    // (a wider panel halves the C read-modify-write passes of the trailing update; its longer factorisation
    // chain only stays hidden behind the update while the trailing matrix is large)
    // matrices too short for look-ahead (no trailing update to hide a panel behind): ONE resident kernel per 4096 columns, a workgroup per
    // 128 x 128 block (chain.hip) -- 36 us per 128 columns instead of the ~50 us of three launches, and no update launches in between
    const bool wide = h->panel_chain && h->chain_wide && np < h->lookahead_min;
    std::vector<int64_t> bnd;
    for (int64_t J0 = 0; J0 < np;) {
        bnd.push_back(J0);
        if (wide) { const int64_t w = np - J0 > h->wide_threshold ? h->wide_block_big : h->wide_block; J0 = (J0 + w < np) ? J0 + w : np; continue; }
        // three widths: `outer_block_big` (2048) while the trailing update hides any chain, NB (1024), and `outer_block_small`
        // (512) for the last `small_threshold` rows, where the chain is what the factorisation waits for: a 512-wide panel's
        // update tiles retire twice as often (K = 512), so the chain's many-workgroup kernels find slots sooner (N=8k -4 %,
        // N=12k -3 %, N=20k +-0 with 512 throughout)
        int64_t w = (h->outer_block_big > NB && np - J0 > h->big_threshold) ? h->outer_block_big : NB;
        if (h->outer_block_small > 0 && h->outer_block_small < w && np - J0 <= h->small_threshold) w = h->outer_block_small;
        J0 = (J0 + w < np) ? J0 + w : np;
    }
    bnd.push_back(np);
    const size_t npan = bnd.size() - 1;
    // a switch between the two streams costs ~12 us (event wait): below ~6k rows the panels are too short to pay for it
    // (measured: N=4000 2.78 ms with, 2.68 without; N=8000 7.48 / 7.58; N=12000 16.4 / 16.9)
    const bool la = h->lookahead && npan > 2 && np >= h->lookahead_min;
    h->chain_alone = la ? 0 : 1;            // (chain.hip: no trailing update runs beside the panel kernels of this factorisation)
    const bool can_split = la && h->cols_split && h->panel_chain && np - bnd[1] >= h->panel_chain_min && chain_streams_concurrent(h) == 1;
    if (!la) {
        for (size_t J = 0; J < npan; ++J) {
            rc = panel_factor_any(h, A, n, np, lda, bnd[J], bnd[J + 1]); if (rc) return rc;
            if (np > bnd[J + 1]) { rc = timed_update(bnd[J], bnd[J + 1], bnd[J + 1], np); if (rc) return rc; }
        }
    } else {
        rc = fvgp_ensure_side(h); if (rc) return rc;
        hipStream_t mainS = h->stream, sideS = h->side;
        rc = panel_factor_any(h, A, n, np, lda, bnd[0], bnd[1]); if (rc) return rc;      // panel 0 on the main stream
        for (size_t J = 0; J + 1 < npan; ++J) {
            const int64_t J0 = bnd[J], Jend = bnd[J + 1], Nend = bnd[J + 2];           // next panel = [Jend, Nend)
            // (1) main: bring the next panel's block columns up to date with panel J
            // `cols_split`: while few rows remain (the chain is what the factorisation waits for) only the next panel's SQUARE is
            // updated before its chain starts; the rows below it follow on the main stream beside the chain, whose block rows
            // below the square wait for a flag in memory that a one-thread kernel raises behind that update (chain.hip) -- the
            // update of (rows below) x (panel) leaves the critical path: one launch + one stream hand-over per panel
            const bool split = can_split && np - Jend >= h->panel_chain_min && np - Jend <= h->cols_split_rows && np > Nend &&
                               (Nend - Jend) / TILE <= FVGP_CHAIN_MAX_BLOCKS;
            unsigned long long cols_tag = 0;
            if (split) {
                GemmDesc s{};          // rows and columns [Jend, Nend): lower tiles
                s.a_kmajor = 0; s.b_nmajor = 0; s.lower = 1; s.M = Nend - Jend; s.N = Nend - Jend; s.K = Jend - J0;
                s.alpha = -1.0; s.beta = 1.0; s.role = 1;
                s.A = A + Jend * lda + J0; s.lda = lda; s.B = s.A; s.ldb = lda; s.C = A + Jend * lda + Jend; s.ldc = lda;
                rc = launch_gemm(h, s); if (rc) return rc;
            } else {
                rc = timed_update(J0, Jend, Jend, Nend); if (rc) return rc;
            }
            HIPCHK(hipEventRecord(h->ev_cols, mainS));
            // (2) side: factor the next panel as soon as (1) is done ...
            HIPCHK(hipStreamWaitEvent(sideS, h->ev_cols, 0));
            h->stream = sideS;
            if (split) { rc = launch_panel_chain(h, A, n, np, lda, Jend, Nend, &cols_tag); }
            else rc = panel_factor_any(h, A, n, np, lda, Jend, Nend);
            h->stream = mainS;
            if (rc) return rc;
            HIPCHK(hipEventRecord(h->ev_panel, sideS));
            if (split) {
                GemmDesc s{};          // rows [Nend, np) x columns [Jend, Nend): every tile
                s.a_kmajor = 0; s.b_nmajor = 0; s.lower = 0; s.M = np - Nend; s.N = Nend - Jend; s.K = Jend - J0;
                s.alpha = -1.0; s.beta = 1.0; s.role = 1;
                s.A = A + Nend * lda + J0; s.lda = lda; s.B = A + Jend * lda + J0; s.ldb = lda; s.C = A + Nend * lda + Jend; s.ldc = lda;
                rc = launch_gemm(h, s); if (rc) return rc;
                rc = launch_chain_cols_ready(h, cols_tag); if (rc) return rc;
            }
            // (3) ... while main applies panel J to everything right of the next panel
            if (np > Nend) { rc = timed_update(J0, Jend, Nend, np); if (rc) return rc; }
            // the next iteration's updates use panel J+1: wait for its factorisation
            HIPCHK(hipStreamWaitEvent(mainS, h->ev_panel, 0));
        }
    }
    if (h->leaf_tiles || h->panel_chain) {
        // the chain only needed the inverses of the 16 x 16 diagonal tiles; the 128 x 128 block inverses the sweeps, the
        // posterior and POTRI use come from one launch over all blocks (a few tens of microseconds on the whole chip
        // instead of 8 us per block on the chain's critical path)
        if (!skip_inverses) { rc = launch_leaf_inverse_batched(h, A, lda, nblk, h->linv); if (rc) return rc; }
    }
    if (enqueue_only) {          // no host round trip: info stays on the device, nothing is timed
        if (info_dev) HIPCHK(hipMemcpyAsync(info_dev, h->dinfo, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        h->winv_ok = false; h->linv_L = skip_inverses ? nullptr : A; h->linv_n = n; h->linv_ld = lda;
        return 0;
    }
    if (h->profile) { rc = get_event(&e_end); if (rc) return rc; HIPCHK(hipEventRecord(e_end, h->stream)); }
    int *hinfo = reinterpret_cast<int *>(h->hpin + RED_SLOTS - 2);
    HIPCHK(hipMemcpyAsync(hinfo, h->dinfo, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int info = *hinfo;
    if (info == 0x7fffffff) { fvgp_set_error("panel chain: a workgroup waited longer than 3 s for a hand-off and the launch was abandoned"); return 1999; }
    if (info > n) info = 0;   // cannot happen: the padding is an identity block
    if (info_host) *info_host = info;
    h->winv_ok = false; h->linv_L = skip_inverses ? nullptr : A; h->linv_n = n; h->linv_ld = lda;
    if (h->profile) {
        h->prof_launches = 0; h->prof_ms = 0; h->prof_flops = 0; h->prof_bytes = 0;
        for (size_t i = 0; i < h->ev_flops.size(); ++i) {
            if (h->ev_flops[i] <= 0.0) continue;
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->ev[1 + 2 * i], h->ev[2 + 2 * i]));
            h->prof_ms += ms; h->prof_flops += h->ev_flops[i]; h->prof_bytes += h->ev_bytes[i]; h->prof_launches += 1.0;
        }
        float tot = 0.f;
        HIPCHK(hipEventElapsedTime(&tot, e_begin, e_end));
        h->prof_total_ms = tot;
    }
    return 0;
}

// B (np x ldb), nrhs columns: in-place solve, vector path (nrhs <= 8)
static int potrs_vec(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb, bool backward) {
    const int64_t np = pad128(n);
    int rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
    rc = ensure_scratch(h, np); if (rc) return rc;
    const int c = (int)nrhs;
    const int C = c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : 8;
    double *Y = h->vec;
    if (h->fwd_sweep && c == 1) {               // one launch for the whole sweep (B is not touched)
        rc = launch_fwd_sweep(h, L, ldl, np, h->linv, B, ldb, Y); if (rc) return rc;
    } else
    for (int64_t k0 = 0; k0 < np; k0 += TILE) {
        rc = launch_fwd_step(h, L, ldl, np, k0, h->linv + (k0 / TILE) * LEAF_DOUBLES, B, ldb, Y, c);
        if (rc) return rc;
    }
    if (!backward) {
        // forward result lives in Y (np x C); copy back to B
        return launch_copy_cols(h, Y, C, B, ldb, np, c, np, c);
    }
    if (h->bwd_sweep && c == 1) return launch_bwd_sweep(h, L, ldl, np, h->linv, Y, B, ldb, c);        // one launch for the whole sweep
    for (int64_t k0 = np - TILE; k0 >= 0; k0 -= TILE) {
        rc = launch_bwd_step(h, L, ldl, np, k0, h->linv + (k0 / TILE) * LEAF_DOUBLES, Y, B, ldb, c);
        if (rc) return rc;
    }
    (void)C;
    return 0;
}

// B (np x ldb), nrhs (multiple of 128) columns: forward block substitution on MFMA GEMMs, two block sizes like
// the factorisation: 128-row steps inside an outer block of `outer_block` rows (updates confined to that block,
// K = 128), then ONE update of everything below with K = outer_block -- the read-modify-write passes over B
// drop by outer_block/128.
static int trsm_fwd_gemm(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t ncols, int64_t ldb) {
    const int64_t np = pad128(n), NB = h->outer_block;
    int rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
    for (int64_t J0 = 0; J0 < np; J0 += NB) {
        const int64_t Jend = (J0 + NB < np) ? J0 + NB : np;
        for (int64_t k0 = J0; k0 < Jend; k0 += TILE) {
            GemmDesc d{};   // X_k = inv(L_kk) B_k
            d.a_kmajor = 0; d.b_nmajor = 1; d.lower = 0; d.M = TILE; d.N = ncols; d.K = TILE; d.alpha = 1.0; d.beta = 0.0;
            d.A = h->linv + (k0 / TILE) * LEAF_DOUBLES; d.lda = TILE;
            d.B = B + k0 * ldb; d.ldb = ldb; d.C = B + k0 * ldb; d.ldc = ldb;
            rc = launch_gemm(h, d); if (rc) return rc;
            const int64_t r0 = k0 + TILE, R = Jend - r0;
            if (R <= 0) continue;
            GemmDesc u{};   // rest of the outer block: B[r0:Jend] -= L[r0:Jend, k] X_k
            u.a_kmajor = 0; u.b_nmajor = 1; u.lower = 0; u.M = R; u.N = ncols; u.K = TILE; u.alpha = -1.0; u.beta = 1.0;
            u.A = L + r0 * ldl + k0; u.lda = ldl; u.B = B + k0 * ldb; u.ldb = ldb; u.C = B + r0 * ldb; u.ldc = ldb;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        if (np > Jend) {
            GemmDesc u{};   // everything below: B[Jend:] -= L[Jend:, J0:Jend] X[J0:Jend]
            u.a_kmajor = 0; u.b_nmajor = 1; u.lower = 0; u.M = np - Jend; u.N = ncols; u.K = Jend - J0; u.alpha = -1.0; u.beta = 1.0;
            u.A = L + Jend * ldl + J0; u.lda = ldl; u.B = B + J0 * ldb; u.ldb = ldb; u.C = B + Jend * ldb; u.ldc = ldb;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
    }
    return 0;
}

// The same substitution on the TRANSPOSED right-hand sides: BT (rows x np, row-major, rows a multiple of 128) holds B^T and
// leaves (L^-1 B)^T.  Every product is then the (M,K) x (N,K) layout of the factorisation's own panel TRSM and trailing
// update -- X_k^T = B_k^T inv(L_kk)^T in place, BT[:, block] -= X^T L[block, k]^T -- i.e. the kernels with the 16-byte
// fragment reads, and what follows (V^T V, row sums) reads contiguous rows.
// The block itself is then ONE product with the inverse of its NB x NB diagonal block (ensure_winv) instead of NB / 128
// steps of two latency-bound launches each.  NB = 2048 up to 1024 rows, where the sweep is a chain of dependent launches and
// half as many are worth the larger block products (N = 20k: P = 8 .. 64 2.53 -> 1.68 ms, 600 5.8 -> 5.4, 1000 8.4 -> 8.1);
// 1024 beyond (flop-bound: P = 2000 / 4000 +0.7 % with 2048).
// LEFT-looking over the outer blocks: block J first receives everything to its left in one product,
//     BT[:, J] -= BT[:, 0:J0] L[J, 0:J0]^T          (rows/128 x NB/128 output tiles, K = J0),
// with K split over enough workgroups to fill the chip (deterministic two-pass reduction).  A right-looking sweep has
// (rows/128) x (remaining blocks) tiles of K = NB per step instead: 1192, 1128, .. tiles on 512 slots lose a quarter of
// the time to partly filled rounds (measured at N = 20k, P = 1000: 7.3 ms for 3.9e11 flops); here every launch is one round.
// `slots`: the workgroups one launch should bring (512 = the whole chip; 256 when two halves of the rows run side by side on two
// streams, trsm_fwd_gemm_t below); scratch: trsm_fwd_scratch(rows, slots) doubles.
// workgroups per output tile of a launch with fewer tiles than slots (split K): s slices take ceil(tiles s / slots) / s rounds of the
// unsplit tile's time; the floor slots / tiles leaves up to a third of the chip idle (192 tiles: 384 of 512), a larger s in two
// rounds can beat it (192 tiles x 5 = 960: 0.4 instead of 0.5).  A small charge per slice for the partial sums' traffic.
static int64_t fill_split(int64_t tiles, int64_t slots) {
    if (tiles >= slots) return 1;
    int64_t best = slots / tiles;
    double cost = 1.0 / (double)best + 0.012 * (double)best;
    for (int64_t sp = best + 1; sp <= 8; ++sp) {
        const double c = (double)((tiles * sp + slots - 1) / slots) / (double)sp + 0.012 * (double)sp;
        if (c < cost - 1e-9) { cost = c; best = sp; }
    }
    return best;
}

static int64_t trsm_fwd_scratch(int64_t rows, int64_t slots, int64_t NB) {
    const int64_t tiles = (rows / TILE) * (NB / TILE);
    const int64_t want = fill_split(tiles, slots);
    return rows * NB + want * rows * NB;
}

// one outer block [J0, J0 + NB) of the sweep for `rows` rows of BT
static int trsm_fwd_gemm_t_block(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *BT, int64_t rows, int64_t ldbt,
                                 int64_t slots, double *scratch, int64_t J0, int64_t NB, int64_t WB) {
    const int64_t np = pad128(n);
    const bool winv = h->block_inverses != 0;
    int rc = 0;
    // scratch: tmp (rows x NB: block J with everything to its left applied) and the split-K partials behind it
    const int64_t tiles = (rows / TILE) * (NB / TILE);
    const int64_t want = fill_split(tiles, slots);                       // workgroups per output tile that fill the launch's share of the chip
    const int64_t tmp_d = rows * NB;
    double *tmp = scratch, *ws = scratch + tmp_d;
    {
        const int64_t Jend = (J0 + NB < np) ? J0 + NB : np, w = Jend - J0;
        bool in_tmp = false;
        if (J0 > 0) {
            GemmDesc u{};
            u.a_kmajor = 0; u.b_nmajor = 0; u.lower = 0; u.M = rows; u.N = w; u.K = J0; u.alpha = -1.0; u.beta = 1.0;
            u.A = BT; u.lda = ldbt; u.B = L + J0 * ldl; u.ldb = ldl; u.C = BT + J0; u.ldc = ldbt;
            int64_t split = want;
            const int64_t max_split = J0 / 512 > 0 ? J0 / 512 : 1;          // at least 512 of K per workgroup
            if (split > max_split) split = max_split;
            if (split > 1) {
                u.split = (int)split; u.split_ws = ws;
                if (winv) { u.split_out = tmp; u.split_ldo = w; in_tmp = true; }   // the reduction drops the block where the next product reads it
            }
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        if (winv) {
            if (!in_tmp) { rc = launch_copy_cols(h, BT + J0, ldbt, tmp, w, rows, w, rows, w); if (rc) return rc; }
            GemmDesc d{};   // X_J^T = B_J^T inv(L_JJ)^T
            d.a_kmajor = 0; d.b_nmajor = 0; d.lower = 0; d.M = rows; d.N = w; d.K = w; d.alpha = 1.0; d.beta = 0.0;
            d.A = tmp; d.lda = w; d.B = h->winv + J0 * WB + J0 % WB; d.ldb = WB; d.C = BT + J0; d.ldc = ldbt;      // (NB < WB: a diagonal sub-block of the WB-wide inverses)
            int64_t split = want;
            if (split > w / TILE) split = w / TILE;
            if (split > 1) {
                d.split = (int)split; d.split_ws = ws;
                // inv(L_JJ) is lower triangular: tile column tj of the product stops at K = 128 (tj + 1), 44 % of the flops never
                // issued (the sums are the same bit for bit: the terms left out are products with explicit zeros).  Slices of whole
                // 128-blocks only.  C2 posterior covariance 7.8 -> 7.46 ms.
                d.split_tri = ((w / 16 + split - 1) / split * 16) % 128 == 0;
            }
            return launch_gemm(h, d);
        }
        for (int64_t k0 = J0; k0 < Jend; k0 += TILE) {
            GemmDesc d{};   // X_k^T = B_k^T inv(L_kk)^T, in place (a workgroup owns whole rows)
            d.a_kmajor = 0; d.b_nmajor = 0; d.lower = 0; d.M = rows; d.N = TILE; d.K = TILE; d.alpha = 1.0; d.beta = 0.0;
            d.A = BT + k0; d.lda = ldbt; d.B = h->linv + (k0 / TILE) * LEAF_DOUBLES; d.ldb = TILE; d.C = BT + k0; d.ldc = ldbt;
            rc = launch_gemm(h, d); if (rc) return rc;
            const int64_t r0 = k0 + TILE, R = Jend - r0;
            if (R <= 0) continue;
            GemmDesc u{};   // rest of the outer block: BT[:, r0:Jend] -= X_k^T L[r0:Jend, k]^T
            u.a_kmajor = 0; u.b_nmajor = 0; u.lower = 0; u.M = rows; u.N = R; u.K = TILE; u.alpha = -1.0; u.beta = 1.0;
            u.A = BT + k0; u.lda = ldbt; u.B = L + r0 * ldl + k0; u.ldb = ldl; u.C = BT + r0; u.ldc = ldbt;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
    }
    return 0;
}

// With 512 or more rows (posterior covariance at P >= 512 points) the rows are cut in two halves that run the same sweep side
// by side on the two streams of the handle, each with launches of 256 workgroups: a step of the sweep is three dependent
// launches with two reductions between them (~66 us of fixed cost per block, 10 blocks at N = 20k), and the other half's
// product fills the chip while they run.
static int trsm_fwd_gemm_t(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *BT, int64_t rows, int64_t ldbt, int64_t block = 0) {
    const bool winv = h->block_inverses != 0;
    // up to 1024 points the sweep is a chain of dependent launches: 2048-wide blocks, half as many (`posterior_block`).  The block
    // width is a function of the call alone (number of rows, the option, `block` of the caller): the same call gives the same bits
    // whether it is the first on a factor or the tenth.  The last doubling level of the inverted blocks costs 1.3 ms at N = 20k, once
    // per factor: a posterior pays it on its first call (the sweeps that follow gain 0.4 ms each at P = 1000, 0.85 at P <= 64);
    // fvgp_hip_trsm_lower, whose callers solve once per factor (the new rows of an append), asks for 1024.
    const int64_t WB = block ? block : (rows <= 1024 ? h->posterior_block : 1024);
    const int64_t NB = WB;
    int rc = winv ? ensure_winv(h, L, n, ldl, WB, NB) : ensure_linv(h, L, n, ldl); if (rc) return rc;
    const bool halves = h->posterior_halves && winv && rows >= 512 && rows <= 1024 && rows % 256 == 0;    // (2048 rows: +3 %)
    const int64_t np = pad128(n);
    if (!halves) {
        rc = ensure_scratch(h, (trsm_fwd_scratch(rows, 512, NB) + 7) / 8); if (rc) return rc;
        for (int64_t J0 = 0; J0 < np && !rc; J0 += NB) rc = trsm_fwd_gemm_t_block(h, L, n, ldl, BT, rows, ldbt, 512, h->vec, J0, NB, WB);
        return rc;
    }
    const int64_t r2 = rows / 2, sc = trsm_fwd_scratch(r2, 256, NB);
    rc = ensure_scratch(h, (2 * sc + 7) / 8); if (rc) return rc;
    rc = fvgp_ensure_side(h); if (rc) return rc;
    hipStream_t mainS = h->stream, sideS = h->side;
    HIPCHK(hipEventRecord(h->ev_cols, mainS));
    HIPCHK(hipStreamWaitEvent(sideS, h->ev_cols, 0));
    for (int64_t J0 = 0; J0 < np && !rc; J0 += NB) {          // the two halves are enqueued block by block (a launch costs the host ~17 us)
        rc = trsm_fwd_gemm_t_block(h, L, n, ldl, BT, r2, ldbt, 256, h->vec, J0, NB, WB);
        if (rc) break;
        h->stream = sideS;
        rc = trsm_fwd_gemm_t_block(h, L, n, ldl, BT + r2 * ldbt, r2, ldbt, 256, h->vec + sc, J0, NB, WB);
        h->stream = mainS;
    }
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->ev_panel, sideS));
    HIPCHK(hipStreamWaitEvent(mainS, h->ev_panel, 0));
    return 0;
}

// backward half, same two block sizes, from the last outer block to the first
static int trsm_bwd_gemm(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t ncols, int64_t ldb) {
    const int64_t np = pad128(n), NB = h->outer_block;
    int rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
    const int64_t npan = (np + NB - 1) / NB;
    for (int64_t J = npan - 1; J >= 0; --J) {
        const int64_t J0 = J * NB, Jend = (J0 + NB < np) ? J0 + NB : np;
        for (int64_t k0 = Jend - TILE; k0 >= J0; k0 -= TILE) {
            GemmDesc d{};   // X_k = inv(L_kk)^T Y_k
            d.a_kmajor = 1; d.b_nmajor = 1; d.lower = 0; d.M = TILE; d.N = ncols; d.K = TILE; d.alpha = 1.0; d.beta = 0.0;
            d.A = h->linv + (k0 / TILE) * LEAF_DOUBLES; d.lda = TILE;
            d.B = B + k0 * ldb; d.ldb = ldb; d.C = B + k0 * ldb; d.ldc = ldb;
            rc = launch_gemm(h, d); if (rc) return rc;
            if (k0 == J0) continue;
            GemmDesc u{};   // rest of the outer block: Y[J0:k0] -= L[k, J0:k0]^T X_k
            u.a_kmajor = 1; u.b_nmajor = 1; u.lower = 0; u.M = k0 - J0; u.N = ncols; u.K = TILE; u.alpha = -1.0; u.beta = 1.0;
            u.A = L + k0 * ldl + J0; u.lda = ldl; u.B = B + k0 * ldb; u.ldb = ldb; u.C = B + J0 * ldb; u.ldc = ldb;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        if (J0 > 0) {
            GemmDesc u{};   // everything above: Y[0:J0] -= L[J0:Jend, 0:J0]^T X[J0:Jend]
            u.a_kmajor = 1; u.b_nmajor = 1; u.lower = 0; u.M = J0; u.N = ncols; u.K = Jend - J0; u.alpha = -1.0; u.beta = 1.0;
            u.A = L + J0 * ldl; u.lda = ldl; u.B = B + J0 * ldb; u.ldb = ldb; u.C = B; u.ldc = ldb;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
    }
    return 0;
}

int fvgp_read_back(fvgp_handle *h, const double *dev, double *host, int count) {
    HIPCHK(hipMemcpyAsync(h->hpin, dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < count; ++i) host[i] = h->hpin[i];
    return fvgp_ipc_check(h);       // (direct collectives: a poll that gave up left stale data behind it -- never hand that to the host)
}

// g_i = 1/2 sum_jk (W_jk - b_j b_k) dK_jk/dtheta_i over the lower triangle of the symmetric W (b may be null):
// one fused pass that re-evaluates dK/dtheta in registers, per-tile partial sums reduced on the host in a fixed order
static int grad_trace_host(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                           const double *W, int64_t ldw, const double *b, int64_t ldb, double *partial, double *grad_host,
                           int64_t col0 = 0, int64_t ncols = 0) {
    GradDesc g{};
    g.col0 = col0; g.ncols = ncols;
    int rc = kmat_desc_from_theta(kernel_id, d, theta, ntheta, &g.k); if (rc) return rc;
    g.k.x1 = x; g.k.n1 = n; g.k.x2 = x; g.k.n2 = n;
    g.kernel_id = kernel_id;
    const bool iso = kernel_id >= 3;
    const int nk = iso ? 2 : d + 1;     // kernel-owned hyperparameters; the rest get a zero gradient
    g.ntheta = nk;
    g.W = W; g.ldw = ldw; g.b = b; g.ldb = ldb;
    g.partial = partial;
    int nblocks = 0;
    rc = launch_grad_trace(h, g, &nblocks); if (rc) return rc;
    // nblocks <= ~80k doubles per theta
    std::vector<double> part((size_t)nblocks * nk);
    HIPCHK(hipMemcpyAsync(part.data(), partial, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < ntheta; ++i) grad_host[i] = 0.0;
    for (int i = 0; i < nk; ++i) {
        long double s = 0.0L;
        for (int bb = 0; bb < nblocks; ++bb) s += part[(size_t)bb * nk + i];
        grad_host[i] = 0.5 * (double)s;
    }
    return 0;
}

extern "C" {

int fvgp_hip_kmat(fvgp_handle *h, int kernel_id, const double *x1, int64_t n1, const double *x2, int64_t n2,
                  int d, const double *theta, int ntheta, const double *vdiag, double *K, int64_t ldk, int uplo, int pad) {
    if (!h) return -1;
    if (!x1) return -3;
    if (n1 <= 0) return -4;
    if (!x2) return -5;
    if (n2 <= 0) return -6;
    if (!theta) return -8;
    if (!K) return -11;
    if (ldk < (pad ? pad128(n2) : n2)) { fvgp_set_error("ldk too small"); return -12; }
    if (uplo != FVGP_FULL && uplo != FVGP_LOWER) return -13;
    if (pad < 0 || pad > 2) return -14;
    HIPCHK(hipSetDevice(h->device));
    KmatDesc k{};
    int rc = kmat_desc_from_theta(kernel_id, d, theta, ntheta, &k);
    if (rc) return rc;
    k.x1 = x1; k.n1 = n1; k.x2 = x2; k.n2 = n2; k.vdiag = vdiag; k.K = K; k.ldk = ldk; k.uplo = uplo; k.pad = pad;
    return launch_kmat(h, k);
}

int fvgp_hip_potrf(fvgp_handle *h, double *A, int64_t n, int64_t lda, int *info_host) {
    if (!h) return -1;
    int rc = check_square(A, n, lda, 2, 3, 4);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    rc = launch_pad_identity(h, A, n, pad128(n), lda);
    if (rc) return rc;
    return potrf_driver(h, A, n, lda, info_host);
}

int fvgp_hip_potrf_dev(fvgp_handle *h, double *A, int64_t n, int64_t lda, int64_t n_logdet, int *info_dev, double *logdet_dev) {
    if (!h) return -1;
    int rc = check_square(A, n, lda, 2, 3, 4);
    if (rc) return rc;
    if (n_logdet < 0 || n_logdet > n) return -5;
    if (!info_dev) return -6;
    HIPCHK(hipSetDevice(h->device));
    rc = launch_pad_identity(h, A, n, pad128(n), lda);
    if (rc) return rc;
    rc = potrf_driver(h, A, n, lda, nullptr, info_dev, true);
    if (rc) return rc;
    if (logdet_dev && n_logdet > 0) return launch_diag_logsum(h, A, n_logdet, lda, logdet_dev);
    return 0;
}

int fvgp_hip_panel_potrf_dev(fvgp_handle *h, double *T, int64_t w, int64_t rows, int64_t ldt, int64_t n_valid,
                             int *info_dev, double *logdet_dev) {
    if (!h) return -1;
    if (!T) return -2;
    if (w <= 0 || w % TILE) { fvgp_set_error("panel_potrf_dev: the panel width must be a positive multiple of 128"); return -3; }
    if (rows < w || rows % TILE) return -4;
    if (ldt < w || (ldt & 1) || ((uintptr_t)T & 15)) return -5;
    if (n_valid < 0 || n_valid > w) return -6;
    if (!info_dev) return -7;
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_blocks(h, w / TILE);
    if (rc) return rc;
    h->winv_ok = false; h->linv_L = nullptr;
    HIPCHK(hipMemsetAsync(h->dinfo, 0, sizeof(int), h->stream));
    // the row-sharded driver's stacked panel: ONE resident kernel with a workgroup per block (8-rank emulation at N = 50 000: 93.2 ms
    // against 96.0 with the launch-per-step chain; with a workgroup per block ROW below the square, panel_chain = 2, 100.6)
    if (h->panel_chain >= 1 && w / TILE <= FVGP_CHAIN_MAX_BLOCKS && rows / TILE <= 1024) { h->chain_alone = h->panel_chain == 2 ? 0 : 2; rc = launch_panel_chain(h, T, n_valid, rows, ldt, 0, w); }      // (2: a workgroup per block, but NOT alone: the rank's trailing update runs beside it)
    else rc = panel_factor_nested(h, T, n_valid, rows, ldt, 0, w);   // leaf / TRSM of every row below / in-panel update per 128 columns, in sub-panels of `inner_block`
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(info_dev, h->dinfo, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    if (logdet_dev && n_valid > 0) return launch_diag_logsum(h, T, n_valid, ldt, logdet_dev);
    return 0;
}

int fvgp_hip_potrs(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    if (!B) return -5;
    if (nrhs <= 0) return -6;
    if (ldb < nrhs) return -7;
    HIPCHK(hipSetDevice(h->device));
    const int64_t np = pad128(n);
    if (np > n) { rc = launch_copy_cols(h, B, ldb, B + n * ldb, ldb, 0, 0, np - n, nrhs); if (rc) return rc; }
    if (nrhs <= FVGP_MAX_RHS_VEC) return potrs_vec(h, L, n, ldl, B, nrhs, ldb, true);
    if (nrhs % 128 || (ldb & 1) || ((uintptr_t)B & 15)) { fvgp_set_error("potrs with nrhs > 8 needs nrhs % 128 == 0, even ldb, 16-byte aligned B"); return -6; }
    rc = trsm_fwd_gemm(h, L, n, ldl, B, nrhs, ldb);
    if (rc) return rc;
    return trsm_bwd_gemm(h, L, n, ldl, B, nrhs, ldb);
}

int fvgp_hip_trsm_lower(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    if (!B) return -5;
    if (nrhs <= 0) return -6;
    if (ldb < nrhs) return -7;
    HIPCHK(hipSetDevice(h->device));
    const int64_t np = pad128(n);
    if (np > n) { rc = launch_copy_cols(h, B, ldb, B + n * ldb, ldb, 0, 0, np - n, nrhs); if (rc) return rc; }
    if (nrhs <= FVGP_MAX_RHS_VEC) return potrs_vec(h, L, n, ldl, B, nrhs, ldb, false);
    if (nrhs % 128 || (ldb & 1) || ((uintptr_t)B & 15)) { fvgp_set_error("trsm with nrhs > 8 needs nrhs % 128 == 0, even ldb, 16-byte aligned B"); return -6; }
    if (h->block_inverses && nrhs <= 1024 && np >= 2048) {
        // few columns against a long factor (the new rows of an append, gp_lin_alg.py:1310-1477; the callables' posterior): the
        // posterior's block sweep on the TRANSPOSED right-hand sides (N / 1024 steps with inverted diagonal blocks instead of
        // N / 128 steps of two latency-bound launches: append of 4 points at N = 20k 10.0 -> 7 ms), two transposes around it
        const size_t need = (size_t)nrhs * np;
        if (need > h->tr_ws_cap) {
            if (h->tr_ws) HIPCHK(hipFree(h->tr_ws));
            h->tr_ws = nullptr; h->tr_ws_cap = 0;
            HIPCHK(hipMalloc((void **)&h->tr_ws, need * sizeof(double)));
            h->tr_ws_cap = need;
        }
        rc = launch_transpose(h, B, ldb, h->tr_ws, np, np, nrhs); if (rc) return rc;
        rc = trsm_fwd_gemm_t(h, L, n, ldl, h->tr_ws, nrhs, np, 1024); if (rc) return rc;
        return launch_transpose(h, h->tr_ws, np, B, ldb, nrhs, np);
    }
    return trsm_fwd_gemm(h, L, n, ldl, B, nrhs, ldb);
}

int fvgp_hip_logdet(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *out_host) {
    if (!h) return -1;
    if (!L) return -2;
    if (n <= 0) return -3;
    if (ldl < n) return -4;
    if (!out_host) return -5;
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_diag_logsum(h, L, n, ldl, h->red);
    if (rc) return rc;
    return fvgp_read_back(h, h->red, out_host, 1);
}

// POTRI on the factorisation's own product layout.  Every product of dtrtri and of W^T W is arranged as (M,K) x (N,K) --
// both operands k-minor, the layout of the trailing update and of its K loop (16-byte swizzled fragment reads, LDS-DMA
// staging, no vector-ALU work) -- by keeping transposes where the textbook schedule reads an operand k-major:
//   dtrtri, 1024-wide panels from the bottom-right corner, W_JJ from the doubled block inverses (ensure_winv):
//        X^T  = W_JJ^T L_2J^T            A = W_JJ^T (transposed copy of the block), B = L_2J          -> work[J, 2]
//        W_2J = -W_22 X                  A = W_22 (k <= row), B = X^T                                 -> over L_2J
//   W^T W = (W^T)(W^T)^T with W^T written into `work` (upper tiles, diagonal tiles transposed), the result straight into L.
// Against the round-2 schedule ((K,N) and (K,M) operands on the 8-byte fragment reads, 14 latency-bound launches per panel
// for W_JJ, ragged K in 438 launches): the same N^3 2/3 flops on the faster kernel in 3 launches per panel.
static int potri_kminor(fvgp_handle *h, double *L, int64_t n, int64_t ldl, double *work, int64_t ldw) {
    const int64_t np = pad128(n), WB = 1024;
    int rc = ensure_winv(h, L, n, ldl); if (rc) return rc;
    rc = ensure_scratch(h, (WB * WB + 7) / 8); if (rc) return rc;
    double *Ujj = h->vec;                                   // W_JJ^T of the panel at hand
    const int64_t npan = (np + WB - 1) / WB;
    for (int64_t J = npan - 1; J >= 0; --J) {
        const int64_t J0 = J * WB, Jend = (J0 + WB < np) ? J0 + WB : np, w = Jend - J0, R = np - Jend;
        const double *Wjj = h->winv + J0 * WB;
        if (R > 0) {
            rc = launch_transpose_lower_tiles(h, Wjj, WB, Ujj, WB, w); if (rc) return rc;
            double *XT = work + J0 * ldw + Jend;            // w x R, in the (free) upper part of work
            GemmDesc t{};   // X^T = W_JJ^T L_2J^T   (W_JJ^T upper: k >= row tile)
            t.a_kmajor = 0; t.b_nmajor = 0; t.lower = 0; t.M = w; t.N = R; t.K = w; t.alpha = 1.0; t.beta = 0.0;
            t.A = Ujj; t.lda = WB; t.B = L + Jend * ldl + J0; t.ldb = ldl; t.C = XT; t.ldc = ldw;
            t.kb0 = 0; t.kbi = TILE; t.kbj = 0; t.ke0 = -1;
            rc = launch_gemm(h, t); if (rc) return rc;
            GemmDesc u{};   // W_2J = -W_22 X   (W_22 lower: k <= row tile), over L_2J
            u.a_kmajor = 0; u.b_nmajor = 0; u.lower = 0; u.M = R; u.N = w; u.K = R; u.alpha = -1.0; u.beta = 0.0;
            u.A = L + Jend * ldl + Jend; u.lda = ldl; u.B = XT; u.ldb = ldw; u.C = L + Jend * ldl + J0; u.ldc = ldl;
            u.kb0 = 0; u.ke0 = TILE; u.kei = TILE; u.kej = 0;
            u.rev_m = 1;    // K grows with the row tile: the long rows start first
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        rc = launch_copy_lower_tiles(h, Wjj, WB, L + J0 * ldl + J0, ldl, w); if (rc) return rc;
    }
    rc = launch_transpose_lower_tiles(h, L, ldl, work, ldw, np); if (rc) return rc;
    GemmDesc s{};   // KV^-1 = W^T W = (W^T)(W^T)^T, lower tiles, k >= row tile
    s.a_kmajor = 0; s.b_nmajor = 0; s.lower = 1; s.M = np; s.N = np; s.K = np; s.alpha = 1.0; s.beta = 0.0;
    s.A = work; s.lda = ldw; s.B = work; s.ldb = ldw; s.C = L; s.ldc = ldl;
    s.kb0 = 0; s.kbi = TILE; s.kbj = 0; s.ke0 = -1;
    rc = launch_gemm(h, s); if (rc) return rc;
    h->winv_ok = false; h->linv_L = nullptr;   // L is gone
    return 0;
}

int fvgp_hip_potri(fvgp_handle *h, double *L, int64_t n, int64_t ldl, double *work, int64_t ldw) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    rc = check_square(work, n, ldw, 5, 3, 6);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (h->potri_kminor) return potri_kminor(h, L, n, ldl, work, ldw);
    const int64_t np = pad128(n);
    const int64_t NB = h->outer_block;
    rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
    // ---- W = inv(L), written over L panel by panel from the bottom-right corner (dtrtri, lower):
    //        W_JJ  = inv(L_JJ)                               (block rows of 128 from the leaf inverses)
    //        W_2J  = -W_22 * (L_2J * W_JJ)                   (two large GEMMs per panel)
    //      `work` holds W_JJ and the intermediate L_2J * W_JJ.
    const int64_t npan = (np + NB - 1) / NB;
    for (int64_t J = npan - 1; J >= 0; --J) {
        const int64_t J0 = J * NB, Jend = (J0 + NB < np) ? J0 + NB : np, w = Jend - J0;
        double *Wjj = work + J0 * ldw + J0;
        const double *Ljj = L + J0 * ldl + J0;
        for (int64_t i0 = 0; i0 < w; i0 += TILE) {
            const double *li = h->linv + ((J0 + i0) / TILE) * LEAF_DOUBLES;
            rc = launch_copy_cols(h, li, TILE, Wjj + i0 * ldw + i0, ldw, TILE, TILE, TILE, TILE); if (rc) return rc;
            if (i0 == 0) continue;
            GemmDesc a{};   // T = L_JJ[i][0:i] * W_JJ[0:i][0:i]   (W lower-triangular: k starts at the column tile)
            a.a_kmajor = 0; a.b_nmajor = 1; a.lower = 0; a.M = TILE; a.N = i0; a.K = i0; a.alpha = 1.0; a.beta = 0.0;
            a.A = Ljj + i0 * ldl; a.lda = ldl; a.B = Wjj; a.ldb = ldw; a.C = Wjj + i0 * ldw; a.ldc = ldw;
            a.kb0 = 0; a.kbi = 0; a.kbj = TILE; a.ke0 = -1;
            rc = launch_gemm(h, a); if (rc) return rc;
            GemmDesc b{};   // W_JJ[i][0:i] = -inv(L_ii) * T  (in place: each tile reads only its own columns)
            b.a_kmajor = 0; b.b_nmajor = 1; b.lower = 0; b.M = TILE; b.N = i0; b.K = TILE; b.alpha = -1.0; b.beta = 0.0;
            b.A = li; b.lda = TILE; b.B = Wjj + i0 * ldw; b.ldb = ldw; b.C = Wjj + i0 * ldw; b.ldc = ldw;
            rc = launch_gemm(h, b); if (rc) return rc;
        }
        const int64_t R = np - Jend;
        if (R > 0) {
            GemmDesc t{};   // T = L_2J * W_JJ -> work   (W_JJ lower: k >= column tile)
            t.a_kmajor = 0; t.b_nmajor = 1; t.lower = 0; t.M = R; t.N = w; t.K = w; t.alpha = 1.0; t.beta = 0.0;
            t.A = L + Jend * ldl + J0; t.lda = ldl; t.B = Wjj; t.ldb = ldw; t.C = work + Jend * ldw + J0; t.ldc = ldw;
            t.kb0 = 0; t.kbi = 0; t.kbj = TILE; t.ke0 = -1;
            rc = launch_gemm(h, t); if (rc) return rc;
            GemmDesc u{};   // W_2J = -W_22 * T -> over L_2J   (W_22 lower: k <= row tile)
            u.a_kmajor = 0; u.b_nmajor = 1; u.lower = 0; u.M = R; u.N = w; u.K = R; u.alpha = -1.0; u.beta = 0.0;
            u.A = L + Jend * ldl + Jend; u.lda = ldl; u.B = work + Jend * ldw + J0; u.ldb = ldw; u.C = L + Jend * ldl + J0; u.ldc = ldl;
            u.kb0 = 0; u.ke0 = TILE; u.kei = TILE; u.kej = 0;
            u.rev_m = 1;    // K grows with the row tile: start the long rows first so the launch has no long tail
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        // W_JJ over L_JJ (its 128-tiles above the block diagonal are never read)
        rc = launch_copy_lower_tiles(h, Wjj, ldw, L + J0 * ldl + J0, ldl, w); if (rc) return rc;
    }
    // ---- KV^-1 = W^T W, lower tiles, k >= row tile; into work, then back over L
    GemmDesc s{};
    s.a_kmajor = 1; s.b_nmajor = 1; s.lower = 1; s.M = np; s.N = np; s.K = np; s.alpha = 1.0; s.beta = 0.0;
    s.A = L; s.lda = ldl; s.B = L; s.ldb = ldl; s.C = work; s.ldc = ldw;
    s.kb0 = 0; s.kbi = TILE; s.kbj = 0; s.ke0 = -1;
    rc = launch_gemm(h, s); if (rc) return rc;
    rc = launch_copy_lower_tiles(h, work, ldw, L, ldl, np); if (rc) return rc;
    h->winv_ok = false; h->linv_L = nullptr;   // L is gone
    return 0;
}

int fvgp_hip_loglik(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                    const double *theta, int ntheta, const double *vdiag, const double *ymean, int ncol,
                    double *KV, int64_t ld, double *alpha, double *out_host, int *info_host) {
    // the contract of this entry: KV holds padded_dim(n) rows, whatever its leading dimension; nothing below them is touched
    const int rc = fvgp_hip_loglik_rows(h, kernel_id, x, n, d, theta, ntheta, vdiag, ymean, ncol, KV, pad128(n), ld, alpha, out_host, info_host);
    return rc <= -13 && rc > -100 ? rc + 1 : rc;        // argument numbers of THIS signature (kv_rows is argument 12 there)
}

int fvgp_hip_loglik_rows(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                         const double *theta, int ntheta, const double *vdiag, const double *ymean, int ncol,
                         double *KV, int64_t kv_rows, int64_t ld, double *alpha, double *out_host, int *info_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!vdiag) { fvgp_set_error("loglik needs the noise variances (vdiag)"); return -8; }
    if (!ymean) return -9;
    if (ncol < 1 || ncol > FVGP_MAX_RHS_VEC) { fvgp_set_error("1 <= ncol <= 8"); return -10; }
    int rc = check_square(KV, n, ld, 11, 4, 13);
    if (rc) return rc;
    if (kv_rows < pad128(n)) { fvgp_set_error("loglik: the scratch needs at least padded_dim(n) rows"); return -12; }
    if (!out_host) return -15;
    HIPCHK(hipSetDevice(h->device));
    const int64_t np = pad128(n);
    KmatDesc k{};
    rc = kmat_desc_from_theta(kernel_id, d, theta, ntheta, &k); if (rc) return rc;
    k.x1 = x; k.n1 = n; k.x2 = x; k.n2 = n; k.vdiag = vdiag; k.K = KV; k.ldk = ld; k.uplo = FVGP_LOWER; k.pad = 1;
    if (h->profile) {
        for (auto &e : h->ev_stage) if (!e) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipEventRecord(h->ev_stage[0], h->stream));
    }
    rc = launch_kmat(h, k); if (rc) return rc;
    if (h->profile) HIPCHK(hipEventRecord(h->ev_stage[1], h->stream));
    // forward solve fused into the factorisation: (y-m)^T is appended as rows n..n+ncol-1 of the padded
    // matrix (diagonal entry large enough to keep the block PD); the panel TRSM / trailing updates then
    // leave z^T = (L^-1 (y-m))^T in those rows and quad = |z|^2.  Needs ncol free padding rows: where padded_dim(n) leaves
    // fewer (n a multiple of 128), the rows go into one more block row -- if the caller SAYS its scratch has it
    // (kv_rows and ld >= fvgp_hip_loglik_dim(n, ncol); never inferred from the leading dimension: a pitched buffer or a row slice
    // of a larger arena holds padded_dim(n) rows only); else the forward solve is a sweep of its own after the factorisation.
    const bool room = (np - n) >= ncol;
    const int64_t npf = room ? np : pad128(n + ncol);
    const bool fused = room || (kv_rows >= npf && ld >= npf);
    if (fused) {
        if (npf > np) { rc = launch_pad_identity(h, KV, np, npf, ld); if (rc) return rc; }
        rc = launch_rhs_rows(h, KV, n, ld, ymean, ncol, vdiag); if (rc) return rc;
    }
    // ONE host round trip per evaluation: the factorisation is only enqueued, its info word comes back with the scalars at the end
    // (what follows a failed factorisation computes on garbage and is thrown away; the profile option times the factorisation with
    // events and keeps the round trip in the middle)
    int info = 0;
    const bool defer = !h->profile;
    const int64_t npd = fused ? npf : 0;
    const bool own_inverses = fused && (h->leaf_tiles || h->panel_chain);      // (see below: the block inverses wait until the appended rows are out again)
    if (defer) { rc = potrf_driver(h, KV, n, ld, nullptr, nullptr, true, npd, own_inverses); if (rc) return rc; }
    else {
        rc = potrf_driver(h, KV, n, ld, &info, nullptr, false, npd, own_inverses); if (rc) return rc;
        if (info_host) *info_host = info;
        if (info != 0) { out_host[0] = out_host[1] = out_host[2] = NAN; return 0; }
    }
    if (h->profile) HIPCHK(hipEventRecord(h->ev_stage[2], h->stream));
    if (fused) {
        // ONE launch: sum log L_ii from the leaves' 1 / L_ii (1 on padding rows), |z|^2 of the appended rows, z (rows of L) -> the
        // (np x C) vector layout of the backward sweep, alpha <- 0
        const int C = ncol <= 1 ? 1 : ncol <= 2 ? 2 : ncol <= 4 ? 4 : 8;
        if (alpha) { rc = ensure_scratch(h, np); if (rc) return rc; }
        rc = launch_loglik_tail(h, h->logdet_parts, npf, KV, ld, n, ncol, h->red, alpha ? h->vec : nullptr, C, np, alpha); if (rc) return rc;
        // hand back the clean factor of blockdiag(K+V, I): identity padding rows again; the 128 x 128 block inverses the sweeps, the
        // posterior and POTRI take are computed from THAT (one batched launch; the last diagonal block without the appended rows)
        rc = launch_pad_identity(h, KV, n, npf, ld); if (rc) return rc;
        if (own_inverses) {
            rc = launch_leaf_inverse_batched(h, KV, ld, npf / TILE, h->linv); if (rc) return rc;
            h->linv_L = KV; h->linv_n = n; h->linv_ld = ld;
        } else if (room) {
            rc = launch_leaf(h, KV + (np - TILE) * ld + (np - TILE), ld, h->linv + (np / TILE - 1) * LEAF_DOUBLES, nullptr, 0, 0, TILE);
            if (rc) return rc;
        }
        if (alpha) {
            if (h->bwd_sweep && ncol == 1) { rc = launch_bwd_sweep(h, KV, ld, np, h->linv, h->vec, alpha, ncol, ncol); if (rc) return rc; }
            else
            for (int64_t k0 = np - TILE; k0 >= 0; k0 -= TILE) {
                rc = launch_bwd_step(h, KV, ld, np, k0, h->linv + (k0 / TILE) * LEAF_DOUBLES, h->vec, alpha, ncol, ncol);
                if (rc) return rc;
            }
        }
    } else {
        rc = launch_neg_log_sum(h, h->logdet_parts, np, h->red); if (rc) return rc;
        if (!alpha) { fvgp_set_error("loglik without alpha needs ncol free padding rows (n % 128 <= 128 - ncol) or a scratch of fvgp_hip_loglik_dim(n, ncol) rows"); return -14; }
        rc = launch_copy_cols(h, ymean, ncol, alpha, ncol, n, ncol, np, ncol); if (rc) return rc;
        rc = potrs_vec(h, KV, n, ld, alpha, ncol, ncol, true); if (rc) return rc;
        rc = launch_dot_rows(h, ymean, ncol, alpha, ncol, n, ncol, h->red + 1); if (rc) return rc;
    }
    if (h->profile) HIPCHK(hipEventRecord(h->ev_stage[3], h->stream));
    double r[2];
    int *hinfo = reinterpret_cast<int *>(h->hpin + RED_SLOTS - 2);
    if (defer) HIPCHK(hipMemcpyAsync(hinfo, h->dinfo, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    rc = fvgp_read_back(h, h->red, r, 2); if (rc) return rc;
    if (defer) {
        info = *hinfo;
        if (info == 0x7fffffff) { fvgp_set_error("panel chain: a workgroup waited longer than 3 s for a hand-off and the launch was abandoned"); return 1999; }
        if (info > n) info = 0;   // cannot happen: the padding is an identity block
        if (info_host) *info_host = info;
        if (info != 0) { out_host[0] = out_host[1] = out_host[2] = NAN; return 0; }
    }
    if (h->profile) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_stage[0], h->ev_stage[1])); h->prof_kmat_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_stage[2], h->ev_stage[3])); h->prof_tail_ms = ms;
        // lower 128-tiles written once (+ the padded diagonal), x read once
        const double tiles = (double)(np / TILE) * (double)(np / TILE + 1) * 0.5;
        h->prof_kmat_bytes = tiles * TILE * TILE * 8.0 + (double)n * d * 8.0;
    }
    const double logdet = 2.0 * r[0], quad = r[1] / (double)ncol;
    out_host[0] = -0.5 * (quad + logdet + (double)n * log(2.0 * M_PI));
    out_host[1] = logdet;
    out_host[2] = quad;
    return 0;
}

int fvgp_hip_loglik_grad(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                         const double *theta, int ntheta, const double *alpha, int ncol, int component,
                         double *KV, int64_t ld, double *work, int64_t ldw, double *grad_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!alpha) return -8;
    if (ncol < 1) return -9;
    if (component < 0 || component >= ncol) return -10;
    if (!grad_host) return -15;
    HIPCHK(hipSetDevice(h->device));
    int rc = fvgp_hip_potri(h, KV, n, ld, work, ldw);
    if (rc) return rc;
    // inv(L) in `work` is dead by now: reuse it as the partial-sum buffer
    return grad_trace_host(h, kernel_id, x, n, d, theta, ntheta, KV, ld, alpha + component, ncol, work, grad_host);
}

int fvgp_hip_grad_trace(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                        const double *theta, int ntheta, const double *W, int64_t ldw,
                        const double *b, int64_t ldb, double *partial, double *grad_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!W) return -8;
    if (ldw < n || (ldw & 1) || ((uintptr_t)W & 15)) return -9;
    if (b && ldb < 1) return -11;
    if (!partial) return -12;
    if (!grad_host) return -13;
    HIPCHK(hipSetDevice(h->device));
    return grad_trace_host(h, kernel_id, x, n, d, theta, ntheta, W, ldw, b, ldb, partial, grad_host);
}

int fvgp_hip_grad_trace_cols(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                             const double *theta, int ntheta, const double *W, int64_t ldw, int64_t col0, int64_t ncols,
                             const double *b, int64_t ldb, double *partial, double *grad_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!W) return -8;
    if (col0 < 0 || col0 % TILE || col0 >= n) return -10;
    if (ncols <= 0) return -11;
    if (ldw < ncols || (ldw & 1) || ((uintptr_t)W & 15)) return -9;
    if (b && ldb < 1) return -13;
    if (!partial) return -14;
    if (!grad_host) return -15;
    HIPCHK(hipSetDevice(h->device));
    return grad_trace_host(h, kernel_id, x, n, d, theta, ntheta, W, ldw, b, ldb, partial, grad_host, col0, ncols);
}

int fvgp_hip_posterior_prepare(fvgp_handle *h, const double *L, int64_t n, int64_t ldl) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (!h->block_inverses || pad128(n) < 2 * TILE) return 0;
    // what the first fvgp_hip_posterior on this factor would build before its sweep: the inverted diagonal blocks at the width a call
    // with up to 1024 points takes (enqueue only)
    return ensure_winv(h, L, n, ldl, h->posterior_block, h->posterior_block);
}

int fvgp_hip_posterior(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d,
                       const double *theta, int ntheta, const double *L, int64_t ldl,
                       const double *alpha, int ncol, const double *xpred, int64_t P,
                       double *kx, int64_t ldk, double *mean_out, double *var_out, double *S_out, int64_t lds) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    int rc = check_square(L, n, ldl, 8, 4, 9);
    if (rc) return rc;
    if (!alpha) return -10;
    if (ncol < 1 || ncol > 128) return -11;
    if (!xpred) return -12;
    if (P <= 0) return -13;
    const int64_t np = pad128(n), Pp = pad128(P);
    if (!kx || ((uintptr_t)kx & 15)) return -14;
    if (ldk < Pp || (ldk & 1)) return -15;
    if (S_out && (lds < Pp || (lds & 1) || ((uintptr_t)S_out & 15))) return -19;
    HIPCHK(hipSetDevice(h->device));
    KmatDesc k{};
    rc = kmat_desc_from_theta(kernel_id, d, theta, ntheta, &k); if (rc) return rc;
    if (P == 1 && h->fwd_sweep && ncol <= FVGP_MAX_RHS_VEC) {
        // ---- ONE prediction point (gradient-based acquisition optimisers ask for one at a time): the cross covariance is a
        //      column, L^-1 k the one-launch forward sweep (N = 20k: 1.1 ms against 1.7 ms of the block sweep below, and no
        //      inverted blocks to build after a new factor); fixed-order sums throughout
        k.x1 = x; k.n1 = n; k.x2 = xpred; k.n2 = 1; k.vdiag = nullptr; k.K = kx; k.ldk = ldk; k.uplo = FVGP_FULL; k.pad = 2;
        rc = launch_kmat(h, k); if (rc) return rc;
        if (mean_out)
            for (int cc = 0; cc < ncol; ++cc) { rc = launch_dot_rows(h, kx, ldk, alpha + cc, ncol, n, 1, mean_out + cc); if (rc) return rc; }
        if (var_out || S_out) {
            rc = ensure_linv(h, L, n, ldl); if (rc) return rc;
            rc = ensure_scratch(h, np / 8 + 16); if (rc) return rc;
            rc = launch_fwd_sweep(h, L, ldl, np, h->linv, kx, ldk, h->vec); if (rc) return rc;          // h->vec <- L^-1 k
            if (var_out) { rc = launch_rows_sumsq_base(h, h->vec, np, np, 1, k.sig, var_out); if (rc) return rc; }      // sigma^2 - |L^-1 k|^2
            if (S_out) {
                KmatDesc kk = k;
                kk.x1 = xpred; kk.n1 = 1; kk.x2 = xpred; kk.n2 = 1; kk.K = S_out; kk.ldk = lds; kk.uplo = FVGP_FULL; kk.pad = 2;
                rc = launch_kmat(h, kk); if (rc) return rc;
                rc = launch_rows_sumsq_base(h, h->vec, np, np, 1, 0.0, h->red + 4); if (rc) return rc;       // -|L^-1 k|^2
                rc = launch_add_matrix(h, S_out, lds, h->red + 4, 1, 1, 1, 1.0); if (rc) return rc;
            }
        }
        return 0;
    }
    // ---- every product runs on the TRANSPOSED cross covariance k(x_pred, x_data), Pp x np with leading dimension np
    //      in the caller's scratch: the substitution then runs on the factorisation's own (M,K) x (N,K) kernels
    //      (trsm_fwd_gemm_t) and S -= V^T V is A A^T of contiguous rows.  Also for a handful of points: the sweep over
    //      2048-blocks is ten dependent steps at N = 20k, 1.7 ms whatever P <= 64, where per-block vector launches took
    //      4.6 / 9.0 ms at P = 2 / 4
    double *KT = kx;
    k.x1 = xpred; k.n1 = P; k.x2 = x; k.n2 = n; k.vdiag = nullptr; k.K = KT; k.ldk = np; k.uplo = FVGP_FULL; k.pad = 2;
    rc = launch_kmat(h, k); if (rc) return rc;
    if (mean_out && ncol <= FVGP_MAX_RHS_VEC) {
        rc = launch_rows_dot(h, KT, np, alpha, ncol, ncol, n, P, mean_out, ncol); if (rc) return rc;
    } else if (mean_out) {
        // many columns of y: GEMM with alpha widened to 128 columns in the handle scratch;
        // the (Pp x 128) result goes to the tail of the same scratch
        rc = ensure_scratch(h, np * 16 + Pp * 16); if (rc) return rc;
        double *aw = h->vec;
        rc = launch_copy_cols(h, alpha, ncol, aw, 128, np, ncol, np, 128); if (rc) return rc;
        double *mw = h->vec + np * 128;
        GemmDesc g{};
        g.a_kmajor = 0; g.b_nmajor = 1; g.lower = 0; g.M = Pp; g.N = 128; g.K = np; g.alpha = 1.0; g.beta = 0.0;
        g.A = KT; g.lda = np; g.B = aw; g.ldb = 128; g.C = mw; g.ldc = 128;
        rc = launch_gemm(h, g); if (rc) return rc;
        rc = launch_copy_cols(h, mw, 128, mean_out, ncol, P, ncol, P, ncol); if (rc) return rc;
    }
    if (var_out || S_out) {
        rc = trsm_fwd_gemm_t(h, L, n, ldl, KT, Pp, np); if (rc) return rc;             // KT <- (L^-1 k)^T
        if (S_out) {
            KmatDesc kk = k;
            kk.x1 = xpred; kk.n1 = P; kk.x2 = xpred; kk.n2 = P; kk.K = S_out; kk.ldk = lds; kk.uplo = FVGP_FULL; kk.pad = 2;
            rc = launch_kmat(h, kk); if (rc) return rc;
            // S -= V^T V = KT KT^T on the 128-tiles on and below the block diagonal only (S is symmetric: 36 of 64 tiles at
            // 1024 points), the rest mirrored; few output tiles and K = np: split K so that the launch fills the chip once
            GemmDesc g{};
            g.a_kmajor = 0; g.b_nmajor = 0; g.lower = 1; g.M = Pp; g.N = Pp; g.K = np; g.alpha = -1.0; g.beta = 1.0;
            g.A = KT; g.lda = np; g.B = KT; g.ldb = np; g.C = S_out; g.ldc = lds;
            const int64_t tr = Pp / TILE, tiles = tr * (tr + 1) / 2;
            // an XCD (64 workgroup slots) gets ceil(tiles / 8) tiles of every K slice: 36 tiles -> 5 -> 12 slices, not 14
            int64_t split = tiles >= 512 ? 1 : 64 / ((tiles + 7) / 8);
            const int64_t max_split = np / 512 > 0 ? np / 512 : 1;       // at least 512 of K per workgroup
            if (tiles >= 512) {
                // more tiles than slots: unsplit, 528 tiles (4096 points) take TWO rounds of 512 for 1.03 rounds of work; s slices per
                // tile take ceil(tiles s / 512) / s rounds -- the smallest s <= 8 that brings that within 15 % of the work
                double best = (double)((tiles + 511) / 512);
                for (int64_t sp = 2; sp <= 8 && sp <= max_split; ++sp) {
                    const double rounds = (double)((tiles * sp + 511) / 512) / (double)sp;
                    if (rounds < best * 0.97) { best = rounds; split = sp; }
                    if (best <= 1.15 * (double)tiles / 512.0) break;
                }
            }
            if (split > max_split) split = max_split;
            if (split > 1) {
                rc = ensure_scratch(h, (split * Pp * Pp + 7) / 8); if (rc) return rc;
                g.split = (int)split; g.split_ws = h->vec;
            }
            rc = launch_gemm(h, g); if (rc) return rc;
            rc = launch_transpose_lower_tiles(h, S_out, lds, S_out, lds, Pp); if (rc) return rc;
        }
        if (var_out) {
            // v_p = k(x_p,x_p) - |L^-1 k_p|^2 ; stationary kernels: k(x,x) = signal variance
            rc = launch_rows_sumsq_base(h, KT, np, np, P, k.sig, var_out); if (rc) return rc;
        }
    }
    return 0;
}

int fvgp_hip_syrk_rowshard(fvgp_handle *h, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                            const double *B, int64_t ldb, double *C, int64_t ldc, int scale, int off,
                            int b_ranks, int b_blocks, int b_off) {
    if (!h) return -1;
    if (!A) return -5;
    if (!B) return -7;
    if (!C) return -9;
    if (scale < 1) return -11;
    if (b_ranks < 1) return -13;
    if (b_off < 0 || (b_ranks > 1 && b_blocks < 1)) return -14;
    if (b_blocks > 0 && N > 0 && (b_off + N / TILE - 1) / b_ranks >= b_blocks) {
        fvgp_set_error("syrk_rowshard: the tile columns run past the gathered blocks"); return -14;
    }
    HIPCHK(hipSetDevice(h->device));
    GemmDesc g{};
    g.a_kmajor = 0; g.b_nmajor = 0; g.lower = 2; g.lower_scale = scale; g.lower_off = off; g.role = 1;
    g.bc_ranks = b_ranks; g.bc_blocks = b_blocks; g.bc_off = b_off;
    g.M = M; g.N = N; g.K = K; g.alpha = -1.0; g.beta = 1.0;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    // only launches of the kernel the roofline names are timed, and no more than 8192 of them between two get_profile calls
    if (!h->profile || gemm_takes_small_tiles(h, g) || h->rs_used >= 2 * 8192) return launch_gemm(h, g);
    // timed with events on the launch stream; algorithmic flops = the tiles with tj <= ti * scale + off
    while (h->rs_ev.size() < h->rs_used + 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->rs_ev.push_back(e); }
    double tiles = 0.0;
    for (int64_t ti = 0; ti < M / TILE; ++ti) {
        int64_t wdt = ti * scale + off + 1;
        if (wdt > N / TILE) wdt = N / TILE;
        if (wdt > 0) tiles += (double)wdt;
    }
    HIPCHK(hipEventRecord(h->rs_ev[h->rs_used], h->stream));
    int rc = launch_gemm(h, g);
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->rs_ev[h->rs_used + 1], h->stream));
    h->rs_used += 2;
    h->rs_flops.push_back(tiles * 128.0 * 128.0 * 2.0 * (double)K);
    return 0;
}

int fvgp_hip_panel_trsm(fvgp_handle *h, const double *D, int64_t nd, int64_t ldd, double *P, int64_t rows, int64_t ldp) {
    if (!h) return -1;
    int rc = check_square(D, nd, ldd, 2, 3, 4);
    if (rc) return rc;
    if (nd % TILE) { fvgp_set_error("panel_trsm: the diagonal block must be a multiple of 128"); return -3; }
    if (!P) return -5;
    if (rows < 0 || rows % TILE) return -6;
    if (ldp < nd || (ldp & 1) || ((uintptr_t)P & 15)) return -7;
    if (rows == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_linv(h, D, nd, ldd); if (rc) return rc;
    // X = P * L^-T by 128-column blocks:  X_k = (P_k - sum_{j<k} X_j L_kj^T) * inv(L_kk)^T
    for (int64_t k0 = 0; k0 < nd; k0 += TILE) {
        if (k0 > 0) {
            GemmDesc u{};
            u.a_kmajor = 0; u.b_nmajor = 0; u.lower = 0; u.M = rows; u.N = TILE; u.K = k0; u.alpha = -1.0; u.beta = 1.0;
            u.A = P; u.lda = ldp; u.B = D + k0 * ldd; u.ldb = ldd; u.C = P + k0; u.ldc = ldp;
            rc = launch_gemm(h, u); if (rc) return rc;
        }
        GemmDesc t{};
        t.a_kmajor = 0; t.b_nmajor = 0; t.lower = 0; t.M = rows; t.N = TILE; t.K = TILE; t.alpha = 1.0; t.beta = 0.0;
        t.A = P + k0; t.lda = ldp; t.B = h->linv + (k0 / TILE) * LEAF_DOUBLES; t.ldb = TILE; t.C = P + k0; t.ldc = ldp;
        rc = launch_gemm(h, t); if (rc) return rc;
    }
    return 0;
}

int fvgp_hip_trsm_lower_t(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    if (!B) return -5;
    if (nrhs <= 0) return -6;
    if (ldb < nrhs) return -7;
    if (nrhs % 128 || (ldb & 1) || ((uintptr_t)B & 15)) { fvgp_set_error("trsm_lower_t needs nrhs % 128 == 0, even ldb, 16-byte aligned B"); return -6; }
    HIPCHK(hipSetDevice(h->device));
    const int64_t np = pad128(n);
    if (np > n) { rc = launch_copy_cols(h, B, ldb, B + n * ldb, ldb, 0, 0, np - n, nrhs); if (rc) return rc; }
    return trsm_bwd_gemm(h, L, n, ldl, B, nrhs, ldb);
}

int fvgp_hip_gemm(fvgp_handle *h, int a_kmajor, int b_nmajor, int lower, int64_t M, int64_t N, int64_t K,
                  double alpha, const double *A, int64_t lda, const double *B, int64_t ldb,
                  double beta, double *C, int64_t ldc) {
    if (!h) return -1;
    if (!A) return -9;
    if (!B) return -11;
    if (!C) return -14;
    HIPCHK(hipSetDevice(h->device));
    GemmDesc g{};
    g.a_kmajor = a_kmajor; g.b_nmajor = b_nmajor; g.lower = lower; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    // few output tiles and a long K (the Schur complement of an append, c - v^T v with K = N: ONE tile walking 4096 of K took
    // 564 us; k^T KV^-1 k of the callables' posterior): K split over workgroups, partials added in a fixed order (handle scratch).
    // An XCD (64 slots) gets ceil(tiles / 8) tiles of every slice; at least 256 of K per slice.
    if (K >= 1024 && M > 0 && N > 0 && M % TILE == 0 && N % TILE == 0 && !(ldc & 1) && !((uintptr_t)C & 15)) {      // (shorter K: the small-tile kernels)
        const int64_t tm = M / TILE, tn = N / TILE, tiles = lower == 1 ? tm * (tm + 1) / 2 : tm * tn;
        int64_t split = tiles >= 256 ? 1 : 64 / ((tiles + 7) / 8);
        if (split > K / 256) split = K / 256;
        if (split > 1 && lower != 2) {
            int rc = ensure_scratch(h, (split * M * N + 7) / 8); if (rc) return rc;
            g.split = (int)split; g.split_ws = h->vec;
        }
    }
    return launch_gemm(h, g);
}

int fvgp_hip_mfma_selftest(fvgp_handle *h, const double *A, const double *B, double *D) {
    if (!h) return -1;
    HIPCHK(hipSetDevice(h->device));
    return launch_mfma_selftest(h, A, B, D);
}

int64_t fvgp_hip_debug_tile_map(int tiles_m, int tiles_n, int lower, int scale, int off, int *out_ti, int *out_tj, int64_t cap) {
    if (tiles_m < 1 || tiles_n < 1 || !out_ti || !out_tj) return -1;
    if (lower < 0 || lower > 2 || (lower == 2 && scale < 1)) return -3;
    return gemm_debug_tile_map(tiles_m, tiles_n, lower, scale, off, out_ti, out_tj, cap);
}

int64_t fvgp_hip_debug_tile_table(int tiles_m, int tiles_n, int lower, int scale, int off, int *out, int64_t cap) {
    if (tiles_m < 1 || tiles_n < 1 || tiles_m >= 32768 || tiles_n >= 32768 || !out) return -1;
    if (lower < 0 || lower > 2 || (lower == 2 && scale < 1)) return -3;
    return gemm_debug_tile_table(tiles_m, tiles_n, lower, scale, off, out, cap);
}

int fvgp_hip_mfma_peak(fvgp_handle *h, double *out, int blocks, int iters) {
    if (!h) return -1;
    if (!out) return -2;
    if (blocks < 1 || iters < 1) return -3;
    HIPCHK(hipSetDevice(h->device));
    return launch_mfma_peak(h, out, blocks, iters);
}

int fvgp_hip_add_lower(fvgp_handle *h, double *A, int64_t n, int64_t lda, const double *B, int64_t ldb, double alpha) {
    if (!h) return -1;
    if (!A) return -2;
    if (n <= 0) return -3;
    if (lda < n) return -4;
    if (!B) return -5;
    if (ldb < n) return -6;
    HIPCHK(hipSetDevice(h->device));
    return launch_add_lower(h, A, lda, B, ldb, n, alpha);
}

int fvgp_hip_trace_dot(fvgp_handle *h, const double *W, int64_t ldw, const double *D, int64_t ldd, const double *b, int64_t ldb,
                       int64_t n, double *out_host) {
    if (!h) return -1;
    if (!W) return -2;
    if (n <= 0) return -8;
    if (ldw < n) return -3;
    if (!D) return -4;
    if (ldd < n) return -5;
    if (b && ldb < 1) return -7;
    if (!out_host) return -9;
    HIPCHK(hipSetDevice(h->device));
    int nblocks = 0;
    int rc = launch_trace_dot(h, W, ldw, D, ldd, b, ldb, n, h->red + 8, &nblocks); if (rc) return rc;      // <= 2048 partial sums
    rc = launch_sum(h, h->red + 8, nblocks, h->red); if (rc) return rc;
    return fvgp_read_back(h, h->red, out_host, 1);
}

int fvgp_hip_add_matrix(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double alpha) {
    if (!h) return -1;
    if (!A) return -2;
    if (!B) return -4;
    if (rows <= 0) return -6;
    if (cols <= 0 || lda < cols || ldb < cols) return -7;
    HIPCHK(hipSetDevice(h->device));
    return launch_add_matrix(h, A, lda, B, ldb, rows, cols, alpha);
}

int fvgp_hip_dot(fvgp_handle *h, const double *a, int64_t lda, const double *b, int64_t ldb, int64_t n, int c, double *out_host) {
    if (!h) return -1;
    if (!a) return -2;
    if (!b) return -4;
    if (n <= 0) return -6;
    if (c < 1 || lda < c || ldb < c) return -7;
    if (!out_host) return -8;
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_dot_rows(h, a, lda, b, ldb, n, c, h->red); if (rc) return rc;
    return fvgp_read_back(h, h->red, out_host, 1);
}

int fvgp_hip_coldot(fvgp_handle *h, const double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double *out) {
    if (!h) return -1;
    if (!A) return -2;
    if (!B) return -4;
    if (rows <= 0) return -6;
    if (cols <= 0 || lda < cols || ldb < cols) return -7;
    if (!out) return -8;
    HIPCHK(hipSetDevice(h->device));
    return launch_coldot(h, A, lda, B, ldb, rows, cols, out);
}

int fvgp_hip_colsumsq(fvgp_handle *h, const double *V, int64_t rows, int64_t ldv, int64_t ncols, double *out) {
    if (!h) return -1;
    if (!V) return -2;
    if (rows <= 0) return -3;
    if (ncols <= 0 || ldv < ncols) return -4;
    if (!out) return -6;
    HIPCHK(hipSetDevice(h->device));
    return launch_colsumsq(h, V, rows, ldv, ncols, 0.0, out, -1.0);
}

int fvgp_hip_symmetrize(fvgp_handle *h, double *A, int64_t n, int64_t lda) {
    if (!h) return -1;
    if (!A) return -2;
    if (n <= 0) return -3;
    if (lda < n) return -4;
    HIPCHK(hipSetDevice(h->device));
    return launch_symmetrize(h, A, n, lda);
}

}  // extern "C"
