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

}  // namespace fvgp_dist
