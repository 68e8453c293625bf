/*
 * fvgp_cpu_dist.cpp -- CPU twin of the row-sharded entry points of include/fvgp_hip.h.  TEST INFRASTRUCTURE ONLY.
 *
 * fvgp_hip_loglik_dist here is the SAME driver the HIP library runs (fvgp_amd/csrc/dist_driver.h: partition, panel loop,
 * buffer offsets, collective calls) instantiated over host loops, so the sharded path runs under gloo in a container without
 * a GPU: the analogue of the reference's in-process Dask cluster fixture (tests/test_fvgp.py:20), which checks
 * "distributed == dense" (tests/test_fvgp.py:3027-3062,3152-3183).  The product never loads this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fvgp_cpu.h"
#include "../../fvgp_amd/csrc/dist_driver.h"

namespace {

struct CpuBackend {
    fvgp_handle *h;
    int use_chain(bool) { return 0; }          // no streams on the host: the chain runs in program order
    int fork() { return 0; }
    int join() { return 0; }
    int zero(double *p, int64_t count) { memset(p, 0, (size_t)count * sizeof(double)); return 0; }
    int zero2d(double *p, int64_t ld, int64_t rows, int64_t cols) { for (int64_t i = 0; i < rows; ++i) memset(p + i * ld, 0, (size_t)cols * sizeof(double)); return 0; }
    int identity2d(double *p, int64_t ld, int64_t n) { for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < n; ++j) p[i * ld + j] = i == j ? 1.0 : 0.0; return 0; }
    int identity(double *p, int64_t n) { return identity2d(p, n, n); }
    int to_device(double *dst, const double *src, int64_t count) { memmove(dst, src, (size_t)count * sizeof(double)); return 0; }
    int to_host(double *dst, const double *src, int64_t count) { memmove(dst, src, (size_t)count * sizeof(double)); return 0; }
    int copy2d(double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols) {
        for (int64_t i = 0; i < rows; ++i) memmove(dst + i * ldd, src + i * lds, (size_t)cols * sizeof(double));
        return 0;
    }
    int kmat_lower(int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta, const double *vdiag, double *K, int64_t ldk) {
        return fvgp_hip_kmat(h, kernel_id, x, n, x, n, d, theta, ntheta, vdiag, K, ldk, FVGP_LOWER, 1);
    }
    int kmat_rows(int kernel_id, const double *x1, int64_t n1, const double *x2, int64_t n2, int d, const double *theta, int ntheta,
                  double *K, int64_t ldk) {
        return fvgp_hip_kmat(h, kernel_id, x1, n1, x2, n2, d, theta, ntheta, nullptr, K, ldk, FVGP_FULL, 2);
    }
    int diag(double *A, int64_t ld, int64_t nrows, int P, int p, int64_t n, int64_t np, const double *v) {
        for (int64_t i = 0; i < nrows; ++i) {
            const int64_t g = ((i >> 7) * P + p) * 128 + (i & 127);
            if (g < n) A[i * ld + g] += v[g];
            else if (g < np) A[i * ld + g] = 1.0;
        }
        return 0;
    }
    int panel_potrf(double *T, int64_t w, int64_t rows, int64_t ldt, int64_t n_valid, int *info_dev, double *logdet_dev) {
        return fvgp_hip_panel_potrf_dev(h, T, w, rows, ldt, n_valid, info_dev, logdet_dev);
    }
    int syrk(int64_t M, int64_t N, int64_t K, const double *A, int64_t lda, const double *Bm, int64_t ldb, double *C, int64_t ldc,
             int scale, int off, int b_ranks, int b_blocks, int b_off) {
        return fvgp_hip_syrk_rowshard(h, M, N, K, A, lda, Bm, ldb, C, ldc, scale, off, b_ranks, b_blocks, b_off);
    }
    int all_gather(const double *send, double *recv, int64_t count, double) {
        return fvgp_hip_all_gather(h, send, recv, count);
    }
};

}  // namespace

extern "C" {

/* dpotrf on the w x w top block (lower), dtrsm of the rows below: the tall panel of the row-sharded driver */
int fvgp_hip_panel_potrf_dev(fvgp_handle *h, double *T, int64_t w, int64_t rows, int64_t ldt, int64_t n_valid, int *info_dev, double *logdet_dev) {
    if (!h) return -1;
    if (!T) return -2;
    if (w <= 0 || w % FVGP_TILE) return -3;
    if (rows < w || rows % FVGP_TILE) return -4;
    if (ldt < w) return -5;
    if (n_valid < 0 || n_valid > w) return -6;
    if (!info_dev) return -7;
    *info_dev = 0;
    for (int64_t j = 0; j < w; ++j) {
        double s = T[j * ldt + j];
        for (int64_t k = 0; k < j; ++k) s -= T[j * ldt + k] * T[j * ldt + k];
        if (!(s > 0.0)) {
            if (j < n_valid) { *info_dev = (int)(j + 1); return 0; }
            s = 1.0;                                    /* cannot happen: the padding is an identity block */
        }
        const double piv = sqrt(s);
        T[j * ldt + j] = piv;
        for (int64_t i = j + 1; i < rows; ++i) {
            double t = T[i * ldt + j];
            const double *ri = T + i * ldt, *rj = T + j * ldt;
            for (int64_t k = 0; k < j; ++k) t -= ri[k] * rj[k];
            T[i * ldt + j] = t / piv;
        }
    }
    if (logdet_dev && n_valid > 0) {
        double s = 0.0;
        for (int64_t j = 0; j < n_valid; ++j) s += log(fabs(T[j * ldt + j]));
        *logdet_dev = 2.0 * s;
    }
    return 0;
}

/* C[ti][tj] -= A[ti] B[block(tj)]^T on the 128-tiles with tj <= ti * scale + off; B rows in all-gather (block-cyclic) order */
int fvgp_hip_syrk_rowshard(fvgp_handle *h, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda, const double *B, int64_t ldb,
                            double *C, int64_t ldc, int scale, int off, int b_ranks, int b_blocks, int b_off) {
    if (!h) return -1;
    if (!A) return -5;
    if (!B) return -7;
    if (!C) return -9;
    if (scale < 1) return -11;
    if (b_ranks < 1) return -13;
    for (int64_t ti = 0; ti < M / FVGP_TILE; ++ti)
        for (int64_t tj = 0; tj < N / FVGP_TILE; ++tj) {
            if (tj > ti * scale + off) continue;
            const int64_t idx = tj + b_off;
            const int64_t rb = (idx % b_ranks) * b_blocks + idx / b_ranks;
            for (int64_t i = 0; i < FVGP_TILE; ++i) {
                const double *ar = A + (ti * FVGP_TILE + i) * lda;
                double *cr = C + (ti * FVGP_TILE + i) * ldc + tj * FVGP_TILE;
                for (int64_t j = 0; j < FVGP_TILE; ++j) {
                    const double *br = B + (rb * FVGP_TILE + j) * ldb;
                    double s = 0.0;
                    for (int64_t k = 0; k < K; ++k) s += ar[k] * br[k];
                    cr[j] -= s;
                }
            }
        }
    return 0;
}

int fvgp_hip_comm_destroy(fvgp_handle *h) {
    if (!h) return -1;
    memset(&h->coll, 0, sizeof h->coll);
    h->coll_rank = 0; h->coll_nranks = 1;
    return 0;
}

int fvgp_hip_comm_init_callbacks(fvgp_handle *h, const fvgp_collectives *cb, int rank, int nranks) {
    if (!h) return -1;
    if (!cb || !cb->all_gather || !cb->all_reduce_sum) return -2;
    if (nranks < 1) return -4;
    if (rank < 0 || rank >= nranks) return -3;
    h->coll = *cb; h->coll_rank = rank; h->coll_nranks = nranks;
    return 0;
}

int fvgp_hip_all_reduce(fvgp_handle *h, double *buf, int64_t count) {
    if (!h) return -1;
    if (!buf) return -2;
    if (count <= 0) return -3;
    if (!h->coll.all_reduce_sum) return h->coll_nranks == 1 ? 0 : 2003;
    return h->coll.all_reduce_sum(h->coll.ctx, buf, count, nullptr);
}

int fvgp_hip_all_gather(fvgp_handle *h, const double *send, double *recv, int64_t count) {
    if (!h) return -1;
    if (!send) return -2;
    if (!recv) return -3;
    if (count <= 0) return -4;
    if (!h->coll.all_gather) {
        if (h->coll_nranks != 1) return 2003;
        if (recv != send) memmove(recv, send, (size_t)count * sizeof(double));
        return 0;
    }
    return h->coll.all_gather(h->coll.ctx, send, recv, count, nullptr);
}

int fvgp_hip_dist_workspace(const fvgp_dist_desc *d, int64_t *out6) {
    if (!d) return -1;
    if (!out6) return -2;
    if (d->n <= 0 || d->nranks < 1 || d->rank < 0 || d->rank >= d->nranks || d->panel < FVGP_TILE || d->panel % FVGP_TILE) return -1;
    fvgp_dist::workspace(*d, out6);
    return 0;
}

int fvgp_hip_loglik_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta, int ntheta, double *out_host, int *info_host) {
    if (!h) return -1;
    if (!d) return -2;
    if (d->n <= 0 || d->d < 1 || d->d > FVGP_MAX_DIM || d->ncol < 1 || d->ncol > FVGP_TILE) return -2;
    if (d->panel < FVGP_TILE || d->panel % FVGP_TILE) return -2;
    if (d->nranks != h->coll_nranks || d->rank != h->coll_rank) { fvgp_cpu_set_err("loglik_dist: rank / nranks differ from the handle's communicator"); return -2; }
    if (!d->x_all || !d->vdiag || !d->zt || !d->A || !d->info_dev || !d->logdet_dev) return -2;
    const bool general = d->nranks > 1 || d->force_general;
    if (general && (!d->T[0] || !d->T[1] || !d->recv[0] || !d->recv[1] || !d->Dfac || !d->gather)) return -2;
    if (!theta) return -3;
    if (!out_host) return -5;
    const fvgp_dist::Geom g = fvgp_dist::geometry(*d);
    memset(d->info_dev, 0, (size_t)g.npan * sizeof(int));
    memset(d->logdet_dev, 0, (size_t)g.npan * sizeof(double));
    CpuBackend b{h};
    const int rc = fvgp_dist::evaluate(b, *d, theta, ntheta);
    if (rc) return rc;
    int info = 0;
    for (int J = 0; J < g.npan && !info; ++J) if (d->info_dev[J] != 0) info = (int)(g.bnd(J) + d->info_dev[J]);
    if (info_host) *info_host = info;
    if (info != 0) { out_host[0] = out_host[1] = out_host[2] = NAN; return 0; }
    double quad = 0.0, logdet = 0.0;
    for (int c = 0; c < d->ncol; ++c)
        for (int64_t j = 0; j < g.n; ++j) { const double z = d->A[(g.zrow + c) * g.ld + j]; quad += z * z; }
    for (int J = 0; J < g.npan; ++J) logdet += d->logdet_dev[J];
    quad /= (double)d->ncol;
    out_host[0] = -0.5 * (quad + logdet + (double)g.n * log(2.0 * M_PI));
    out_host[1] = logdet;
    out_host[2] = quad;
    return 0;
}

/* after the factorisation: the same drivers as the HIP library's (dist_driver.h), over the host loops of this twin */
int64_t fvgp_hip_dist_scratch(const fvgp_dist_desc *d, int what, int64_t npred, int64_t slab) {
    if (!d || what < 0 || what > 2 || npred < 0 || (what == 2 && (slab < FVGP_TILE || slab % FVGP_TILE))) return -1;
    return fvgp_dist::scratch_doubles(*d, what, npred, slab);
}
int fvgp_hip_solve_dist(fvgp_handle *h, const fvgp_dist_desc *d, double *alpha_out, double *ws) {
    if (!h) return -1;
    if (!d || !d->keep_factor) return -2;
    if (!alpha_out) return -3;
    if (!ws) return -4;
    CpuBackend b{h};
    return fvgp_dist::solve_backward(b, h, *d, alpha_out, ws);
}
int fvgp_hip_posterior_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta, int ntheta, const double *xpred, int64_t npred,
                            const double *k_pre, const double *kk_pre, const double *alpha, double *mean_out, double *S_out, double *ws) {
    if (!h) return -1;
    if (!d || !d->keep_factor) return -2;
    if (!theta) return -3;
    if (npred <= 0) return -6;
    if (!alpha) return -9;
    if (!mean_out) return -10;
    if (!ws) return -12;
    CpuBackend b{h};
    return fvgp_dist::posterior(b, h, *d, theta, ntheta, xpred, npred, k_pre, kk_pre, alpha, mean_out, S_out, ws);
}
int fvgp_hip_grad_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta, int ntheta, const double *alpha, int component,
                       int64_t slab, double *grad_host, double *diag_out, double *ws) {
    if (!h) return -1;
    if (!d || !d->keep_factor) return -2;
    if (!theta) return -3;
    if (!alpha) return -5;
    if (slab < FVGP_TILE || slab % FVGP_TILE) return -7;
    if (!grad_host) return -8;
    if (!ws) return -10;
    CpuBackend b{h};
    return fvgp_dist::gradient(b, h, *d, theta, ntheta, alpha, component, slab, grad_host, diag_out, ws);
}

}  /* extern "C" */
