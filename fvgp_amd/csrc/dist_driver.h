// dist_driver.h -- ONE evaluation of the path on a row-sharded K+V, written once over a backend.
//
// The reference's only parallel decomposition is gp2Scale's block decomposition of the covariance over Dask workers
// (fvgp/gp2Scale_covariance.py:381-396; x broadcast once: gp_prior.py:319-322; switch: gp.py:419-439).  Here the same
// pattern carries the DENSE factorisation: 128-row blocks dealt block-cyclically (block b -> rank b mod P), x replicated,
// right-looking blocked Cholesky with one panel of look-ahead, the forward solve riding along as one more block row.
//
// The driver below sequences that evaluation -- assembly of the rank's rows, per panel: gather of the diagonal block,
// factorisation of the tall panel, all-gather of the panel factor, trailing update of the rank's block rows -- over a
// backend that supplies the operations:
//   * libfvgp_hip.so instantiates it with its HIP kernels, two streams and RCCL (api.hip: fvgp_hip_loglik_dist);
//   * the CPU twin of the ABI (oracle/cpu_abi, test infrastructure) instantiates it with host loops, so the partition
//     and collective logic runs under gloo in the build container, without a GPU -- the analogue of the reference's
//     in-process Dask cluster fixture (tests/test_fvgp.py:20).
// The collectives are function pointers either way (fvgp_collectives): RCCL on the GPU box, Python callbacks in the tests.
#pragma once
#include <stdint.h>
#include "../../include/fvgp_hip.h"

namespace fvgp_dist {

constexpr int64_t T128 = FVGP_TILE;

struct Geom {
    int64_t n, np, nblk, nb_max, nloc, zrow, NB, ld;
    int P, p, npan;
    int64_t bnd(int J) const { const int64_t b = (int64_t)J * NB; return b < np ? b : np; }
};

static inline int64_t ceil_pos(int64_t a, int64_t b) { return a <= 0 ? 0 : (a + b - 1) / b; }     // max(0, ceil(a / b))

static inline Geom geometry(const fvgp_dist_desc &d) {
    Geom g;
    g.n = d.n; g.np = (d.n + T128 - 1) / T128 * T128; g.nblk = g.np / T128;
    g.P = d.nranks; g.p = d.rank; g.NB = d.panel;
    g.nb_max = (g.nblk + g.P - 1) / g.P;          // block rows per rank (uniform, padded)
    g.nloc = g.nb_max + 1;                        // + the block of right-hand-side rows
    g.zrow = g.nb_max * T128;
    g.ld = g.np;
    g.npan = (int)((g.np + g.NB - 1) / g.NB);
    return g;
}

// sizes (doubles) of the caller-owned buffers of fvgp_dist_desc
static inline void workspace(const fvgp_dist_desc &d, int64_t out[6]) {
    const Geom g = geometry(d);
    const bool general = g.P > 1 || d.force_general;
    const int64_t per = ((g.NB / T128) + g.P - 1) / g.P;                 // diagonal-block row blocks a rank contributes at most
    out[0] = g.nloc * T128 * g.np;                                       // A
    out[1] = general ? (g.NB + g.nloc * T128) * g.NB : 0;                // T[0], T[1] each
    out[2] = general ? g.P * g.nb_max * T128 * g.NB : 0;                 // recv[0], recv[1] each
    out[3] = general ? (int64_t)g.npan * g.NB * g.NB : 0;                // Dfac
    out[4] = general ? (1 + g.P) * per * T128 * g.NB : 0;                // gather scratch of the diagonal block (send + recv)
    out[5] = g.npan;                                                     // info_dev (ints), logdet_dev (doubles)
}

// Apply panel J to block columns [c0, c1) of this rank's rows below the panel (lower tiles only).
template <class B>
static int update(B &b, const fvgp_dist_desc &d, const Geom &g, int J, int64_t c0, int64_t c1, const double *low) {
    if (c1 <= c0) return 0;
    const int64_t J0 = g.bnd(J), Jend = g.bnd(J + 1), w = Jend - J0;
    const int64_t b1 = Jend / T128;
    const int64_t l0 = ceil_pos(b1 - g.p, g.P);                          // first local block row below the panel
    const int64_t M = (g.nloc - l0) * T128, cb = c0 / T128;
    const bool general = g.P > 1 || d.force_general;
    double *A = d.A;
    if (general) {
        const int64_t L0 = b1 / g.P, k = (g.nb_max - L0) * T128;
        (void)k;
        return b.syrk(M, c1 - c0, w, low, w, d.recv[J % 2], w, A + l0 * T128 * g.ld + c0, g.ld, g.P, (int)(l0 * g.P + g.p - cb),
                      g.P, (int)(g.nb_max - L0), (int)(cb - L0 * g.P));
    }
    return b.syrk(M, c1 - c0, w, A + l0 * T128 * g.ld + J0, g.ld, A + c0 * g.ld + J0, g.ld, A + l0 * T128 * g.ld + c0, g.ld, 1,
                  (int)(l0 - cb), 1, 0, 0);
}

// Panel J on the chain stream: the diagonal block is gathered from its owners and stacked on top of this rank's rows of the
// panel; the tall panel is factored like a panel of the single-GPU driver (the top block redundantly on every rank -- no pivot
// traffic inside the panel); the solved rows are all-gathered.  *low_out = this rank's rows below the panel, compact (ld = w).
template <class B>
static int chain(B &b, const fvgp_dist_desc &d, const Geom &g, int J, const double **low_out) {
    const int64_t J0 = g.bnd(J), Jend = g.bnd(J + 1), w = Jend - J0;
    const int64_t b0 = J0 / T128, b1 = Jend / T128, nbw = w / T128;
    int64_t n_valid = g.n - J0; if (n_valid < 0) n_valid = 0; if (n_valid > w) n_valid = w;
    int *info = d.info_dev + J; double *ld = d.logdet_dev + J;
    double *A = d.A;
    const bool general = g.P > 1 || d.force_general;
    int rc;
    if (!general) {                                                      // the panel is contiguous in A: in place
        *low_out = nullptr;
        return b.panel_potrf(A + J0 * g.ld + J0, w, g.nloc * T128 - J0, g.ld, n_valid, info, ld);
    }
    const int P = g.P, p = g.p;
    const int64_t la = ceil_pos(b0 - p, P), lb = ceil_pos(b1 - p, P);    // local blocks [la, lb) lie in the panel's rows
    const int64_t L0 = b1 / P;                                           // uniform first gathered local block (L0 <= lb)
    const int64_t kt = (g.nloc - L0) * T128;                             // rows below: local blocks L0.. and the (y-m)^T block
    double *T = d.T[J % 2];
    double *D = T, *low = T + w * w;
    // 1. the diagonal block from its owners: every rank sends its (at most `per`) row blocks of the panel, one all-gather
    const int64_t per = ((g.NB / T128) + P - 1) / P, chunk = per * T128 * w;
    double *S = d.gather, *G = d.gather + per * T128 * g.NB;
    if (lb > la) { rc = b.copy2d(S, w, A + la * T128 * g.ld + J0, g.ld, (lb - la) * T128, w); if (rc) return rc; }
    if (per == 1 && P == nbw && b0 % P == 0) {
        // one block per rank and block t of the panel on rank t: the gathered order IS the panel's row order, the all-gather
        // writes the diagonal block in place (P = 8 ranks, 1024-wide panels)
        rc = b.all_gather(S, D, chunk, 8.0 * (P - 1) * chunk); if (rc) return rc;
    } else {
        rc = b.all_gather(S, G, chunk, 8.0 * (P - 1) * chunk); if (rc) return rc;
        for (int64_t t = 0; t < nbw; ++t) {
            const int64_t gb = b0 + t, q = gb % P, li = gb / P, laq = ceil_pos(b0 - q, P);
            rc = b.copy2d(D + t * T128 * w, w, G + q * chunk + (li - laq) * T128 * w, w, T128, w); if (rc) return rc;
        }
    }
    // 2. this rank's rows at / below the panel, compact
    rc = b.copy2d(low, w, A + L0 * T128 * g.ld + J0, g.ld, kt, w); if (rc) return rc;
    // (a panel narrower than P blocks: on some ranks the uniform first block L0 lies ABOVE the panel, rows whose entries in these
    // columns were never assembled -- nobody reads what the solve makes of them, but they are solved and gathered: zeros, not stale memory)
    if (la > L0) { rc = b.zero(low, (la - L0) * T128 * w); if (rc) return rc; }
    // 3. factor the tall panel
    rc = b.panel_potrf(T, w, w + kt, w, n_valid, info, ld); if (rc) return rc;
    // 4. the factored diagonal block stays replicated for the later solves
    rc = b.copy2d(d.Dfac + (int64_t)J * g.NB * g.NB, g.NB, D, w, w, w); if (rc) return rc;
    const double *mylow = low + (lb - L0) * T128 * w;                    // rows strictly below the panel
    *low_out = mylow;
    if (d.keep_factor) {                                                 // the solves that follow read the factor from A
        for (int64_t l = la; l < lb; ++l) {
            const int64_t t = l * P + p - b0;
            rc = b.copy2d(A + l * T128 * g.ld + J0, g.ld, D + t * T128 * w, w, T128, w); if (rc) return rc;
        }
        rc = b.copy2d(A + lb * T128 * g.ld + J0, g.ld, mylow, w, (g.nloc - lb) * T128, w); if (rc) return rc;
    } else {                                                             // only the (y-m)^T rows are read back at the end
        rc = b.copy2d(A + g.zrow * g.ld + J0, g.ld, low + (g.nb_max - L0) * T128 * w, w, T128, w); if (rc) return rc;
    }
    // 5. the panel factor to every rank (the one large collective: sum ~ 4 N^2 bytes per rank)
    if (Jend < g.np) {
        const int64_t k = (g.nb_max - L0) * T128;
        rc = b.all_gather(low, d.recv[J % 2], k * w, 8.0 * (P - 1) * k * w); if (rc) return rc;
    }
    return 0;
}

// This rank's block rows of K+V (columns up to the block's own diagonal tile: the rest is never read), identity on the
// padding, (y-m)^T in the extra block.
template <class B>
static int assemble(B &b, const fvgp_dist_desc &d, const Geom &g, const double *theta, int ntheta) {
    double *A = d.A;
    int rc;
    if (d.preassembled) {                      // the caller's rows of K are in place: noise / identity on the diagonal only
        rc = b.diag(A, g.ld, g.nb_max * T128, g.P, g.p, g.n, g.np, d.vdiag); if (rc) return rc;
    } else if (g.P == 1) {
        rc = b.kmat_lower(d.kernel_id, d.x_all, g.n, d.d, theta, ntheta, d.vdiag, A, g.ld); if (rc) return rc;
    } else {
        for (int64_t l = 0; l < g.nb_max; ++l) {
            const int64_t gb = l * g.P + g.p, r0 = gb * T128;
            int64_t rows = g.n - r0; if (rows > T128) rows = T128;
            if (rows <= 0) { rc = b.zero(A + l * T128 * g.ld, T128 * g.ld); if (rc) return rc; continue; }
            int64_t cols = (gb + 1) * T128; if (cols > g.n) cols = g.n;
            rc = b.kmat_rows(d.kernel_id, d.x_all + r0 * d.d, rows, d.x_all, cols, d.d, theta, ntheta, A + l * T128 * g.ld, g.ld);
            if (rc) return rc;
        }
        rc = b.diag(A, g.ld, g.nb_max * T128, g.P, g.p, g.n, g.np, d.vdiag); if (rc) return rc;
    }
    return b.copy2d(A + g.zrow * g.ld, g.ld, d.zt, g.np, T128, g.np);
}

// assemble + factor, all enqueued without a host round trip; the appended rows come out as (L^-1 (y-m))^T
template <class B>
static int evaluate(B &b, const fvgp_dist_desc &d, const double *theta, int ntheta) {
    const Geom g = geometry(d);
    b.use_chain(false);
    int rc = assemble(b, d, g, theta, ntheta); if (rc) return rc;
    const double *low[2] = {nullptr, nullptr};
    b.fork();
    b.use_chain(true);
    rc = chain(b, d, g, 0, &low[0]); if (rc) return rc;
    b.use_chain(false);
    for (int J = 0; J + 1 < g.npan; ++J) {
        b.join();                                                        // panel J factored and gathered
        rc = update(b, d, g, J, g.bnd(J + 1), g.bnd(J + 2), low[J % 2]); if (rc) return rc;      // next panel's columns first ...
        b.fork();
        b.use_chain(true);
        rc = chain(b, d, g, J + 1, &low[(J + 1) % 2]); if (rc) return rc;                        // ... so its chain overlaps the rest
        b.use_chain(false);
        rc = update(b, d, g, J, g.bnd(J + 2), g.np, low[J % 2]); if (rc) return rc;
    }
    b.join();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// After the factorisation: what the reference's distributed mode answers through the same object (gp_kv.py:574-593,
// gp_posterior.py:139-288, tests/test_fvgp.py:3112-3149) -- KVinvY, the posterior, the gradient -- on the factor that
// evaluate(keep_factor = 1) left: the rank's block rows of L in A, every factored diagonal block replicated (Dfac).
// The operations are the library's own ABI entries (gemm, triangular solves, the trace pass) on the handle's stream; the
// backend only moves memory.  `ws`: caller-owned scratch, sizes from scratch_doubles().

struct PanelRows { int64_t la, lb, first, J0, Jend; };
// local blocks [la, lb) are this rank's rows of panel J; they sit at positions first, first + P, ... of the panel's 128-row blocks
static inline PanelRows panel_rows(const Geom &g, int J) {
    PanelRows r;
    r.J0 = g.bnd(J); r.Jend = g.bnd(J + 1);
    const int64_t b0 = r.J0 / T128, b1 = r.Jend / T128;
    r.la = ceil_pos(b0 - g.p, g.P);
    r.lb = ceil_pos(b1 - g.p, g.P); if (r.lb < r.la) r.lb = r.la;
    r.first = r.la * g.P + g.p - b0;
    return r;
}
static inline const double *diag_block(const fvgp_dist_desc &d, const Geom &g, int J, int64_t *ld) {
    if (g.P > 1 || d.force_general) { *ld = g.NB; return d.Dfac + (int64_t)J * g.NB * g.NB; }
    *ld = g.ld; return d.A + g.bnd(J) * g.ld + g.bnd(J);
}

enum { SCRATCH_SOLVE = 0, SCRATCH_POSTERIOR = 1, SCRATCH_GRADIENT = 2 };
static inline int64_t pad128_(int64_t n) { return (n + T128 - 1) / T128 * T128; }
static inline int64_t scratch_doubles(const fvgp_dist_desc &d, int what, int64_t npred, int64_t slab) {
    const Geom g = geometry(d);
    const int64_t rows = g.nb_max * T128, pp = pad128_(npred);
    const int64_t solve = 2 * g.np * T128 + 3 * g.NB * T128 + T128 * T128;
    if (what == SCRATCH_SOLVE) return solve;
    if (what == SCRATCH_POSTERIOR) return rows * pp + rows * T128 + g.NB * pp;                       // k, alpha's local rows, the panel's rows of the solve
    const int64_t nt = g.np / T128;
    return rows * g.np + g.np * slab + g.NB * g.np + nt * (slab / T128) * (d.d + 2) + 64;            // inv(L)'s rows, one slab of the Gram matrix, the solve's panel, trace partials
}

// KVinvY = L^-T z (the second half of cho_solve), replicated: alpha (np x 128, row-major; columns >= ncol carry zeros).
// Column sweep over the panels from the last: the rows of panel J are solved against the replicated diagonal block
// (redundantly, no traffic), then every rank adds L[its rows of J, columns left of J]^T alpha_J to its own partial sum; one
// all-reduce of an NB x 128 slice per panel completes the right-hand side of the next.
template <class B>
static int solve_backward(B &b, fvgp_handle *h, const fvgp_dist_desc &d, double *alpha, double *ws) {
    const Geom g = geometry(d);
    double *Y = ws, *S = Y + g.np * T128, *G = S + g.np * T128, *Sg = G + g.NB * T128, *mine = Sg + g.NB * T128, *I = mine + g.NB * T128;
    int rc = b.identity(I, T128); if (rc) return rc;
    // z^T sits in the extra block row of A: Y (np x 128) = its transpose
    rc = fvgp_hip_gemm(h, 1, 1, 0, g.np, T128, T128, 1.0, d.A + g.zrow * g.ld, g.ld, I, T128, 0.0, Y, T128); if (rc) return rc;
    rc = b.zero(S, g.np * T128); if (rc) return rc;
    for (int J = g.npan - 1; J >= 0; --J) {
        const PanelRows r = panel_rows(g, J);
        const int64_t w = r.Jend - r.J0;
        rc = b.copy2d(Sg, T128, S + r.J0 * T128, T128, w, T128); if (rc) return rc;
        rc = fvgp_hip_all_reduce(h, Sg, w * T128); if (rc) return rc;
        rc = b.copy2d(G, T128, Y + r.J0 * T128, T128, w, T128); if (rc) return rc;
        rc = fvgp_hip_add_matrix(h, G, T128, Sg, T128, w, T128, -1.0); if (rc) return rc;      // z_J - the sum of the ranks' parts
        int64_t ldd; const double *D = diag_block(d, g, J, &ldd);
        rc = fvgp_hip_invalidate_factor(h); if (rc) return rc;
        rc = fvgp_hip_trsm_lower_t(h, D, w, ldd, G, T128, T128); if (rc) return rc;
        rc = b.copy2d(alpha + r.J0 * T128, T128, G, T128, w, T128); if (rc) return rc;
        if (r.lb > r.la && r.J0 > 0) {
            for (int64_t l = r.la; l < r.lb; ++l) {
                rc = b.copy2d(mine + (l - r.la) * T128 * T128, T128, G + (r.first + (l - r.la) * g.P) * T128 * T128, T128, T128, T128);
                if (rc) return rc;
            }
            rc = fvgp_hip_gemm(h, 1, 1, 0, r.J0, T128, (r.lb - r.la) * T128, 1.0, d.A + r.la * T128 * g.ld, g.ld, mine, T128, 1.0, S, T128);
            if (rc) return rc;
        }
    }
    return 0;
}

// Bm <- this rank's rows of L^-1 B_global for a right-hand side distributed by rows like the matrix itself (Bm: nb_max*128
// local rows x m columns, m a multiple of 128).  Per panel: the panel's rows are summed to every rank, solved against the
// replicated diagonal block, and applied to the rank's later rows as one GEMM.  triangular: B_global is lower triangular (the
// identity: inv(L)), so panel J only carries its first Jend columns.  Gs: NB x m doubles.
template <class B>
static int forward_trsm(B &b, fvgp_handle *h, const fvgp_dist_desc &d, const Geom &g, double *Bm, int64_t ldb, int64_t m, bool triangular, double *Gs) {
    int rc;
    for (int J = 0; J < g.npan; ++J) {
        const PanelRows r = panel_rows(g, J);
        const int64_t w = r.Jend - r.J0;
        const int64_t mJ = triangular ? (m < r.Jend ? m : r.Jend) : m;
        double *G = Gs;                                                     // w x mJ, contiguous: the collective needs it
        if (g.P > 1) {
            rc = b.zero(G, w * mJ); if (rc) return rc;
            for (int64_t l = r.la; l < r.lb; ++l) {
                rc = b.copy2d(G + (r.first + (l - r.la) * g.P) * T128 * mJ, mJ, Bm + l * T128 * ldb, ldb, T128, mJ); if (rc) return rc;
            }
            rc = fvgp_hip_all_reduce(h, G, w * mJ); if (rc) return rc;
        } else {
            rc = b.copy2d(G, mJ, Bm + r.J0 * ldb, ldb, w, mJ); if (rc) return rc;
        }
        int64_t ldd; const double *D = diag_block(d, g, J, &ldd);
        rc = fvgp_hip_invalidate_factor(h); if (rc) return rc;
        rc = fvgp_hip_trsm_lower(h, D, w, ldd, G, mJ, mJ); if (rc) return rc;
        for (int64_t l = r.la; l < r.lb; ++l) {
            rc = b.copy2d(Bm + l * T128 * ldb, ldb, G + (r.first + (l - r.la) * g.P) * T128 * mJ, mJ, T128, mJ); if (rc) return rc;
        }
        const int64_t below = (g.nb_max - r.lb) * T128;
        if (below > 0) {
            rc = fvgp_hip_gemm(h, 0, 1, 0, below, mJ, w, -1.0, d.A + r.lb * T128 * g.ld + r.J0, g.ld, G, mJ, 1.0, Bm + r.lb * T128 * ldb, ldb);
            if (rc) return rc;
        }
    }
    return 0;
}

// this rank's rows of a matrix whose rows are the global rows: block l <- src block l * P + p (zero beyond the last block)
template <class B>
static int local_rows(B &b, const Geom &g, double *dst, int64_t ldd, const double *src, int64_t lds, int64_t cols) {
    for (int64_t l = 0; l < g.nb_max; ++l) {
        const int64_t gb = l * g.P + g.p;
        const int rc = gb < g.nblk ? b.copy2d(dst + l * T128 * ldd, ldd, src + gb * T128 * lds, lds, T128, cols) : b.zero2d(dst + l * T128 * ldd, ldd, T128, cols);
        if (rc) return rc;
    }
    return 0;
}

// k^T KVinvY and kk - k^T KV^-1 k (gp_posterior.py:139-182,229-288) at the factored hyperparameters: every rank assembles its
// own rows of k(x_data, x_pred) (or takes them from the caller: k_pre, host kernel callables); the mean and V^T V (V = L^-1 k)
// are summed over the ranks.  alpha: KVinvY (np x 128, replicated).  mean_out: pp x 128, S_out: pp x pp or null (both device,
// replicated; pp = padded_dim(npred)); kk_pre: the caller's k(x_pred, x_pred) (pp x pp, only read on rank 0) or null.
template <class B>
static int posterior(B &b, fvgp_handle *h, const fvgp_dist_desc &d, const double *theta, int ntheta, const double *xpred, int64_t npred,
                     const double *k_pre, const double *kk_pre, const double *alpha, double *mean_out, double *S_out, double *ws) {
    const Geom g = geometry(d);
    const int64_t rows = g.nb_max * T128, pp = pad128_(npred);
    double *k = ws, *a_loc = k + rows * pp, *Gs = a_loc + rows * T128;
    int rc;
    if (k_pre) { rc = b.copy2d(k, pp, k_pre, pp, rows, pp); if (rc) return rc; }
    else {
        rc = b.zero(k, rows * pp); if (rc) return rc;
        for (int64_t l = 0; l < g.nb_max; ++l) {
            const int64_t r0 = (l * g.P + g.p) * T128;
            int64_t cnt = g.n - r0; if (cnt > T128) cnt = T128;
            if (cnt <= 0) continue;
            rc = fvgp_hip_kmat(h, d.kernel_id, d.x_all + r0 * d.d, cnt, xpred, npred, d.d, theta, ntheta, nullptr, k + l * T128 * pp, pp, FVGP_FULL, 0);
            if (rc) return rc;
        }
    }
    rc = local_rows(b, g, a_loc, T128, alpha, T128, T128); if (rc) return rc;
    rc = fvgp_hip_gemm(h, 1, 1, 0, pp, T128, rows, 1.0, k, pp, a_loc, T128, 0.0, mean_out, T128); if (rc) return rc;
    rc = fvgp_hip_all_reduce(h, mean_out, pp * T128); if (rc) return rc;
    if (!S_out) return 0;
    rc = forward_trsm(b, h, d, g, k, pp, pp, false, Gs); if (rc) return rc;
    rc = b.zero(S_out, pp * pp); if (rc) return rc;
    if (g.p == 0) {
        if (kk_pre) { rc = b.copy2d(S_out, pp, kk_pre, pp, pp, pp); if (rc) return rc; }
        else { rc = fvgp_hip_kmat(h, d.kernel_id, xpred, npred, xpred, npred, d.d, theta, ntheta, nullptr, S_out, pp, FVGP_FULL, 0); if (rc) return rc; }
    }
    rc = fvgp_hip_gemm(h, 1, 1, 0, pp, pp, rows, -1.0, k, pp, k, pp, 1.0, S_out, pp); if (rc) return rc;
    return fvgp_hip_all_reduce(h, S_out, pp * pp);
}

// 1/2 (tr(KV^-1 dK_i) - b^T dK_i b), b = KVinvY[:, component] (gp_marginal_likelihood.py:262-300) for the kernel-owned
// hyperparameters.  inv(L) is built by rows with the distributed forward solve (N^2 / P doubles per rank); the Gram matrix of
// the rank's rows (the sum over ranks is KV^-1, never formed) is walked in column slabs of `slab` columns, traced by the fused
// pass and dropped; the (ntheta,) partial results are summed over the ranks.  grad_host: ntheta doubles; diag_out: np doubles
// (device, replicated: diag(KV^-1), the gradients of noise-function hyperparameters need it) or null.
template <class B>
static int gradient(B &b, fvgp_handle *h, const fvgp_dist_desc &d, const double *theta, int ntheta, const double *alpha, int component,
                    int64_t slab, double *grad_host, double *diag_out, double *ws) {
    const Geom g = geometry(d);
    const int64_t rows = g.nb_max * T128, np = g.np;
    double *W = ws, *Gm = W + rows * np, *Gs = Gm + np * slab, *partial = Gs + g.NB * np, *gdev = partial + (np / T128) * (slab / T128) * (d.d + 2);
    int rc = b.zero(W, rows * np); if (rc) return rc;
    for (int64_t l = 0; l < g.nb_max; ++l) {                                // this rank's rows of the identity
        const int64_t gb = l * g.P + g.p;
        if (gb < g.nblk) { rc = b.identity2d(W + l * T128 * np + gb * T128, np, T128); if (rc) return rc; }
    }
    rc = forward_trsm(b, h, d, g, W, np, np, true, Gs); if (rc) return rc;
    for (int i = 0; i < ntheta; ++i) grad_host[i] = 0.0;
    double part[FVGP_MAX_DIM + 2];
    for (int64_t c0 = 0; c0 < np; c0 += slab) {
        const int64_t wc = slab < np - c0 ? slab : np - c0;
        // rows >= c0 of the slab: (W^T W)[c0:, c0:c0+wc] = W[:, c0:]^T W[:, c0:c0+wc]
        rc = fvgp_hip_gemm(h, 1, 1, 0, np - c0, wc, rows, 1.0, W + c0, np, W + c0, np, 0.0, Gm + c0 * slab, slab); if (rc) return rc;
        if (c0 < g.n) {
            const int64_t nc = wc < g.n - c0 ? wc : g.n - c0;
            rc = fvgp_hip_grad_trace_cols(h, d.kernel_id, d.x_all, g.n, d.d, theta, ntheta, Gm, slab, c0, nc,
                                          g.p == 0 ? alpha + component : nullptr, T128, partial, part);
            if (rc) return rc;
            for (int i = 0; i < ntheta; ++i) grad_host[i] += part[i];
        }
    }
    rc = b.to_device(gdev, grad_host, ntheta); if (rc) return rc;
    rc = fvgp_hip_all_reduce(h, gdev, ntheta); if (rc) return rc;
    rc = b.to_host(grad_host, gdev, ntheta); if (rc) return rc;
    if (diag_out) {
        rc = fvgp_hip_colsumsq(h, W, rows, np, np, diag_out); if (rc) return rc;
        rc = fvgp_hip_all_reduce(h, diag_out, np); if (rc) return rc;
    }
    return 0;
}

}  // namespace fvgp_dist
