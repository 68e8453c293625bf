/*
 * fvgp_cpu.c -- CPU twin of the C ABI in include/fvgp_hip.h.  TEST INFRASTRUCTURE ONLY (lives under oracle/).
 *
 * The same entry points with the same argument meaning, padding rules, lower-triangle convention and
 * return / info codes as libfvgp_hip.so, in plain scalar C on host pointers.  It exists so that the ABI's
 * semantics can be exercised in a container without a GPU -- in particular DRIVEN BY THE REFERENCE ITSELF
 * through the reference's own plug points (kernel_function, linalg_mode=[f_factor, f_solve, f_logdet];
 * fvgp/gp_kv.py:457-458,552-554,625-628, tests/test_fvgp.py:417-426,5030-5052): tests/test_cpu_abi.py.
 * The product (fvgp_amd/) never loads it: there is no CPU fallback.
 *
 * Each routine restates the reference step it stands for:
 *   kmat        fvgp/kernels.py:16-33,98-118,166-188,440-481; gp_prior.py:376-400; gp_bo.py:115-126; addKV gp_kv.py:639-669
 *   potrf       scipy cho_factor(lower=True) -> dpotrf          gp_lin_alg.py:237-269
 *   potrs       cho_solve -> dpotrs                              gp_lin_alg.py:289-328
 *   logdet      2*sum(log|diag|)                                 gp_lin_alg.py:331-360
 *   loglik      gp_marginal_likelihood.py:137-179, gp_kv.py:574-593
 *   grad        gp_marginal_likelihood.py:224-309 in the trace form 1/2 sum (KVinv - b b^T) o dK
 *   posterior   gp_posterior.py:139-182,229-288
 * The row-sharded evaluation (fvgp_hip_loglik_dist and the collective entries) is in fvgp_cpu_dist.cpp: the SAME driver the
 * HIP library runs (fvgp_amd/csrc/dist_driver.h) over host loops.
 * Not provided (GPU-only scheduling pieces): streams, potrf_dev, MFMA probes.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fvgp_cpu.h"

static char g_err[256] = "";
static void set_err(const char *s) { snprintf(g_err, sizeof g_err, "%s", s); }
void fvgp_cpu_set_err(const char *s) { set_err(s); }
static int64_t pad128(int64_t n) { return (n + FVGP_TILE - 1) / FVGP_TILE * FVGP_TILE; }

int fvgp_hip_version(void) { return 100; }
const char *fvgp_hip_last_error_string(void) { return g_err; }
int64_t fvgp_hip_padded_dim(int64_t n) { return pad128(n); }
int64_t fvgp_hip_loglik_dim(int64_t n, int ncol) { if (n <= 0 || ncol < 1) return -1; return pad128(n) - n >= ncol ? pad128(n) : pad128(n + ncol); }
int64_t fvgp_hip_workspace_bytes(int64_t n, int64_t npred) { return (n <= 0 || npred < 0) ? -1 : 0; }
int fvgp_hip_create(fvgp_handle **out, int device, void *stream) {
    (void)stream;
    if (!out) return -1;
    if (device != 0) { set_err("no such device"); return -2; }
    *out = (fvgp_handle *)calloc(1, sizeof(fvgp_handle));
    (*out)->outer_block = 1024;
    (*out)->coll_nranks = 1;
    return 0;
}
int fvgp_hip_destroy(fvgp_handle *h) { free(h); return 0; }
int fvgp_hip_sync(fvgp_handle *h) { return h ? 0 : -1; }
int fvgp_hip_set_option(fvgp_handle *h, const char *key, int64_t value) { (void)value; return !h ? -1 : (!key ? -2 : 0); }
int fvgp_hip_get_profile(fvgp_handle *h, double *out) { if (!h) return -1; if (!out) return -2; memset(out, 0, 8 * sizeof(double)); return 0; }
int fvgp_hip_invalidate_factor(fvgp_handle *h) { return h ? 0 : -1; }

/* ---- kernels ------------------------------------------------------------------------------------ */
typedef struct { int kind, d, iso; double sig, invl[FVGP_MAX_DIM]; } kdesc;
static int kdesc_from_theta(int kernel_id, int d, const double *theta, int ntheta, kdesc *k) {
    if (kernel_id < 0 || kernel_id > 5) { set_err("unknown kernel id"); return -2; }
    if (d < 1 || d > FVGP_MAX_DIM) { set_err("input dimension out of range"); return -7; }
    k->iso = kernel_id >= 3;
    if (ntheta < (k->iso ? 2 : d + 1)) { set_err("too few hyperparameters for this kernel"); return -9; }
    k->kind = kernel_id % 3; k->d = d; k->sig = theta[0];
    for (int i = 0; i < d; ++i) k->invl[i] = 1.0 / (k->iso ? theta[1] : theta[1 + i]);
    return 0;
}
static double radial(int kind, double r2) {
    if (kind == 0) return exp(-0.5 * r2);                                             /* kernels.py:16-33 with l = 1 */
    const double r = sqrt(r2);
    if (kind == 1) return (1.0 + sqrt(3.0) * r) * exp(-sqrt(3.0) * r);                 /* kernels.py:98-118 */
    return (1.0 + sqrt(5.0) * r + (5.0 / 3.0) * r2) * exp(-sqrt(5.0) * r);             /* kernels.py:166-188 */
}
static double kval(const kdesc *k, const double *a, const double *b) {
    double r2 = 0.0;
    for (int i = 0; i < k->d; ++i) { const double e = (a[i] - b[i]) * k->invl[i]; r2 += e * e; }
    return k->sig * radial(k->kind, r2);
}

int fvgp_hip_kmat(fvgp_handle *h, int kernel_id, const double *x1, int64_t n1, const double *x2, int64_t n2,
                  int d, const double *theta, int ntheta, const double *vdiag, double *K, int64_t ldk, int uplo, int pad) {
    if (!h) return -1;
    if (!x1) return -3;
    if (n1 <= 0) return -4;
    if (!x2) return -5;
    if (n2 <= 0) return -6;
    if (!theta) return -8;
    if (!K) return -11;
    if (ldk < (pad ? pad128(n2) : n2)) { set_err("ldk too small"); return -12; }
    if (uplo != FVGP_FULL && uplo != FVGP_LOWER) return -13;
    if (pad < 0 || pad > 2) return -14;
    kdesc k;
    int rc = kdesc_from_theta(kernel_id, d, theta, ntheta, &k);
    if (rc) return rc;
    const int64_t r1 = pad ? pad128(n1) : n1, c1 = pad ? pad128(n2) : n2;
    for (int64_t i = 0; i < r1; ++i)
        for (int64_t j = 0; j < c1; ++j) {
            if (uplo == FVGP_LOWER && j / FVGP_TILE > i / FVGP_TILE) continue;       /* whole tiles on / below the diagonal */
            double v;
            if (i < n1 && j < n2) {
                v = kval(&k, x1 + i * d, x2 + j * d);
                if (vdiag && i == j) v += vdiag[i];                                    /* addKV fused, gp_kv.py:663-665 */
            } else v = (pad == 1 && i == j) ? 1.0 : 0.0;
            K[i * ldk + j] = v;
        }
    return 0;
}

/* ---- dense Cholesky --------------------------------------------------------------------------------- */
static int check_square(const void *A, int64_t n, int64_t ld, int argA, int argn, int argld) {
    if (!A) return -argA;
    if (n <= 0) return -argn;
    if (ld < pad128(n) || (ld & 1)) { set_err("leading dimension must be even and >= padded_dim(n)"); return -argld; }
    return 0;
}
static void pad_identity(double *A, int64_t n, int64_t lda) {
    for (int64_t i = n; i < pad128(n); ++i)
        for (int64_t j = 0; j <= i; ++j) A[i * lda + j] = (i == j) ? 1.0 : 0.0;
}
/* lower Cholesky in place over rows [0, m); returns dpotrf's info */
static int chol_lower(double *A, int64_t m, int64_t lda) {
    for (int64_t j = 0; j < m; ++j) {
        double s = A[j * lda + j];
        for (int64_t k = 0; k < j; ++k) s -= A[j * lda + k] * A[j * lda + k];
        if (!(s > 0.0)) return (int)(j + 1);
        const double p = sqrt(s);
        A[j * lda + j] = p;
        for (int64_t i = j + 1; i < m; ++i) {
            double t = A[i * lda + j];
            for (int64_t k = 0; k < j; ++k) t -= A[i * lda + k] * A[j * lda + k];
            A[i * lda + j] = t / p;
        }
    }
    return 0;
}
int fvgp_hip_potrf(fvgp_handle *h, double *A, int64_t n, int64_t lda, int *info_host) {
    if (!h) return -1;
    int rc = check_square(A, n, lda, 2, 3, 4);
    if (rc) return rc;
    pad_identity(A, n, lda);
    const int info = chol_lower(A, n, lda);
    if (info_host) *info_host = info;
    return 0;
}
static void fwd_solve(const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    for (int64_t c = 0; c < nrhs; ++c)
        for (int64_t i = 0; i < n; ++i) {
            double s = B[i * ldb + c];
            for (int64_t k = 0; k < i; ++k) s -= L[i * ldl + k] * B[k * ldb + c];
            B[i * ldb + c] = s / L[i * ldl + i];
        }
}
static void bwd_solve(const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    for (int64_t c = 0; c < nrhs; ++c)
        for (int64_t i = n - 1; i >= 0; --i) {
            double s = B[i * ldb + c];
            for (int64_t k = i + 1; k < n; ++k) s -= L[k * ldl + i] * B[k * ldb + c];
            B[i * ldb + c] = s / L[i * ldl + i];
        }
}
static int check_rhs(const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    if (!B) return -5;
    if (nrhs <= 0) return -6;
    if (ldb < nrhs) return -7;
    for (int64_t i = n; i < pad128(n); ++i) for (int64_t c = 0; c < nrhs; ++c) B[i * ldb + c] = 0.0;
    return 0;
}
int fvgp_hip_potrs(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_rhs(L, n, ldl, B, nrhs, ldb);
    if (rc) return rc;
    fwd_solve(L, n, ldl, B, nrhs, ldb);
    bwd_solve(L, n, ldl, B, nrhs, ldb);
    return 0;
}
int fvgp_hip_trsm_lower(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_rhs(L, n, ldl, B, nrhs, ldb);
    if (rc) return rc;
    fwd_solve(L, n, ldl, B, nrhs, ldb);
    return 0;
}
int fvgp_hip_trsm_lower_t(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *B, int64_t nrhs, int64_t ldb) {
    if (!h) return -1;
    int rc = check_rhs(L, n, ldl, B, nrhs, ldb);
    if (rc) return rc;
    bwd_solve(L, n, ldl, B, nrhs, ldb);
    return 0;
}
int fvgp_hip_logdet(fvgp_handle *h, const double *L, int64_t n, int64_t ldl, double *out_host) {
    if (!h) return -1;
    if (!L) return -2;
    if (n <= 0) return -3;
    if (ldl < n) return -4;
    if (!out_host) return -5;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += log(fabs(L[i * ldl + i]));
    *out_host = 2.0 * s;
    return 0;
}
/* KV^-1 (lower) over L: columns of the inverse by two triangular solves */
int fvgp_hip_potri(fvgp_handle *h, double *L, int64_t n, int64_t ldl, double *work, int64_t ldw) {
    if (!h) return -1;
    int rc = check_square(L, n, ldl, 2, 3, 4);
    if (rc) return rc;
    rc = check_square(work, n, ldw, 5, 3, 6);
    if (rc) return rc;
    double *col = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t j = 0; j < n; ++j) {
        for (int64_t i = 0; i < n; ++i) col[i] = (i == j) ? 1.0 : 0.0;
        fwd_solve(L, n, ldl, col, 1, 1);
        bwd_solve(L, n, ldl, col, 1, 1);
        for (int64_t i = j; i < n; ++i) work[i * ldw + j] = col[i];
    }
    free(col);
    for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j <= i; ++j) L[i * ldl + j] = work[i * ldw + j];
    return 0;
}

/* ---- gradient trace ------------------------------------------------------------------------------------- */
static int grad_trace(int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                      const double *W, int64_t ldw, const double *b, int64_t ldb, double *grad) {
    kdesc k;
    int rc = kdesc_from_theta(kernel_id, d, theta, ntheta, &k);
    if (rc) return rc;
    const int nk = k.iso ? 2 : d + 1;
    for (int i = 0; i < ntheta; ++i) grad[i] = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            const double wt = (i == j ? 1.0 : 2.0) * (W[i * ldw + j] - (b ? b[i * ldb] * b[j * ldb] : 0.0));
            double e2[FVGP_MAX_DIM], r2 = 0.0;
            for (int q = 0; q < d; ++q) { const double e = (x[i * d + q] - x[j * d + q]) * k.invl[q]; e2[q] = e * e; r2 += e2[q]; }
            const double r = sqrt(r2);
            double phi = radial(k.kind, r2), cf;
            if (k.kind == 0) cf = k.sig * phi;                                               /* dk/dl_i = k D_i^2 / l_i^3 */
            else if (k.kind == 1) cf = 3.0 * k.sig * exp(-sqrt(3.0) * r);                     /* gp_prior.py:421-436 */
            else cf = (5.0 / 3.0) * k.sig * (1.0 + sqrt(5.0) * r) * exp(-sqrt(5.0) * r);      /* gp_bo.py:167-201 */
            grad[0] += wt * phi;
            for (int q = 0; q < d; ++q) grad[k.iso ? 1 : 1 + q] += wt * cf * e2[q] * k.invl[q];
        }
    for (int i = 0; i < nk; ++i) grad[i] *= 0.5;
    return 0;
}
int fvgp_hip_grad_trace(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                        const double *W, int64_t ldw, const double *b, int64_t ldb, double *partial, double *grad_host) {
    (void)partial;
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!W) return -8;
    if (ldw < n) return -9;
    if (!grad_host) return -13;
    return grad_trace(kernel_id, x, n, d, theta, ntheta, W, ldw, b, ldb, grad_host);
}

/* ---- fused evaluations -------------------------------------------------------------------------------------- */
int fvgp_hip_loglik(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                    const double *vdiag, const double *ymean, int ncol, double *KV, int64_t ld, double *alpha,
                    double *out_host, int *info_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!vdiag) { set_err("loglik needs the noise variances (vdiag)"); return -8; }
    if (!ymean) return -9;
    if (ncol < 1 || ncol > FVGP_MAX_RHS_VEC) { set_err("1 <= ncol <= 8"); return -10; }
    int rc = check_square(KV, n, ld, 11, 4, 12);
    if (rc) return rc;
    if (!alpha) return -13;
    if (!out_host) return -14;
    rc = fvgp_hip_kmat(h, kernel_id, x, n, x, n, d, theta, ntheta, vdiag, KV, ld, FVGP_LOWER, 1);
    if (rc) return rc;
    int info = 0;
    rc = fvgp_hip_potrf(h, KV, n, ld, &info);
    if (rc) return rc;
    if (info_host) *info_host = info;
    if (info) { out_host[0] = out_host[1] = out_host[2] = NAN; return 0; }
    for (int64_t i = 0; i < pad128(n); ++i) for (int c = 0; c < ncol; ++c) alpha[i * ncol + c] = i < n ? ymean[i * ncol + c] : 0.0;
    fwd_solve(KV, n, ld, alpha, ncol, ncol);
    bwd_solve(KV, n, ld, alpha, ncol, ncol);
    double quad = 0.0, logdet = 0.0;
    for (int64_t i = 0; i < n; ++i) for (int c = 0; c < ncol; ++c) quad += ymean[i * ncol + c] * alpha[i * ncol + c];
    quad /= (double)ncol;                                                       /* gp_marginal_likelihood.py:171: / ncols */
    fvgp_hip_logdet(h, KV, n, ld, &logdet);
    out_host[0] = -0.5 * (quad + logdet + (double)n * log(2.0 * M_PI));
    out_host[1] = logdet;
    out_host[2] = quad;
    return 0;
}
/* the rows the caller owns below padded_dim(n) are not needed on the host (no fused forward solve here): same results */
int fvgp_hip_loglik_rows(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                         const double *vdiag, const double *ymean, int ncol, double *KV, int64_t kv_rows, int64_t ld, double *alpha,
                         double *out_host, int *info_host) {
    if (KV && n > 0 && kv_rows < pad128(n)) { set_err("loglik: the scratch needs at least padded_dim(n) rows"); return -12; }
    const int rc = fvgp_hip_loglik(h, kernel_id, x, n, d, theta, ntheta, vdiag, ymean, ncol, KV, ld, alpha, out_host, info_host);
    return rc <= -12 && rc > -100 ? rc - 1 : rc;
}
int fvgp_hip_loglik_grad(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                         const double *alpha, int ncol, int component, double *KV, int64_t ld, double *work, int64_t ldw,
                         double *grad_host) {
    if (!h) return -1;
    if (!x) return -3;
    if (n <= 0) return -4;
    if (!theta) return -6;
    if (!alpha) return -8;
    if (ncol < 1) return -9;
    if (component < 0 || component >= ncol) return -10;
    if (!grad_host) return -15;
    int rc = fvgp_hip_potri(h, KV, n, ld, work, ldw);
    if (rc) return rc;
    return grad_trace(kernel_id, x, n, d, theta, ntheta, KV, ld, alpha + component, ncol, grad_host);
}
int fvgp_hip_posterior(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                       const double *L, int64_t ldl, const double *alpha, int ncol, const double *xpred, int64_t P,
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
    if (!kx) return -14;
    if (ldk < pad128(P) || (ldk & 1)) return -15;
    if (S_out && lds < pad128(P)) return -19;
    kdesc k;
    rc = kdesc_from_theta(kernel_id, d, theta, ntheta, &k);
    if (rc) return rc;
    rc = fvgp_hip_kmat(h, kernel_id, x, n, xpred, P, d, theta, ntheta, NULL, kx, ldk, FVGP_FULL, 2);   /* gp_prior.py:200-215 */
    if (rc) return rc;
    if (mean_out)
        for (int64_t p = 0; p < P; ++p) for (int c = 0; c < ncol; ++c) {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += kx[i * ldk + p] * alpha[i * ncol + c];              /* gp_posterior.py:158 */
            mean_out[p * ncol + c] = s;
        }
    if (var_out || S_out) {
        fwd_solve(L, n, ldl, kx, P, ldk);                                                              /* gp_posterior.py:120-136 */
        if (S_out) {
            rc = fvgp_hip_kmat(h, kernel_id, xpred, P, xpred, P, d, theta, ntheta, NULL, S_out, lds, FVGP_FULL, 2);
            if (rc) return rc;
            for (int64_t p = 0; p < P; ++p) for (int64_t q = 0; q < P; ++q) {
                double s = 0.0;
                for (int64_t i = 0; i < n; ++i) s += kx[i * ldk + p] * kx[i * ldk + q];
                S_out[p * lds + q] -= s;
            }
        }
        if (var_out)
            for (int64_t p = 0; p < P; ++p) {
                double s = 0.0;
                for (int64_t i = 0; i < n; ++i) s += kx[i * ldk + p] * kx[i * ldk + p];
                var_out[p] = k.sig - s;
            }
    }
    return 0;
}

/* ---- building blocks ---------------------------------------------------------------------------------------- */
int fvgp_hip_gemm(fvgp_handle *h, int a_kmajor, int b_nmajor, int lower, int64_t M, int64_t N, int64_t K,
                  double alpha, const double *A, int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc) {
    if (!h) return -1;
    if (!A) return -9;
    if (!B) return -11;
    if (!C) return -14;
    if (M % 128 || N % 128 || K % 16 || K < 0) { set_err("gemm: M,N must be multiples of 128 and K of 16"); return -5; }
    /* row of C at a time, k outermost inside it: contiguous rows of B^T / columns handled by the inner loop (the row-sharded
     * gradient's Gram slabs run through here in the gloo tests) */
    double *acc = (double *)malloc((size_t)N * sizeof(double));
    if (!acc) return -1;
    for (int64_t i = 0; i < M; ++i) {
        const int64_t jmax = lower ? ((i / 128 + 1) * 128 < N ? (i / 128 + 1) * 128 : N) : N;
        for (int64_t j = 0; j < jmax; ++j) acc[j] = 0.0;
        if (b_nmajor) {
            for (int64_t k = 0; k < K; ++k) {
                const double a = a_kmajor ? A[k * lda + i] : A[i * lda + k];
                const double *br = B + k * ldb;
                for (int64_t j = 0; j < jmax; ++j) acc[j] += a * br[j];
            }
        } else {
            for (int64_t j = 0; j < jmax; ++j) {
                const double *br = B + j * ldb;
                double t = 0.0;
                if (a_kmajor) for (int64_t k = 0; k < K; ++k) t += A[k * lda + i] * br[k];
                else { const double *ar = A + i * lda; for (int64_t k = 0; k < K; ++k) t += ar[k] * br[k]; }
                acc[j] = t;
            }
        }
        for (int64_t j = 0; j < jmax; ++j) C[i * ldc + j] = alpha * acc[j] + (beta != 0.0 ? beta * C[i * ldc + j] : 0.0);
    }
    free(acc);
    return 0;
}
int fvgp_hip_add_matrix(fvgp_handle *h, double *A, int64_t lda, const double *B, int64_t ldb, int64_t rows, int64_t cols, double alpha) {
    if (!h) return -1;
    if (!A) return -2;
    if (!B) return -4;
    for (int64_t i = 0; i < rows; ++i) for (int64_t j = 0; j < cols; ++j) A[i * lda + j] += alpha * B[i * ldb + j];
    return 0;
}
int fvgp_hip_colsumsq(fvgp_handle *h, const double *V, int64_t rows, int64_t ldv, int64_t ncols, double *out) {
    if (!h) return -1;
    if (!V) return -2;
    if (!out) return -6;
    for (int64_t j = 0; j < ncols; ++j) out[j] = 0.0;
    for (int64_t i = 0; i < rows; ++i) for (int64_t j = 0; j < ncols; ++j) out[j] += V[i * ldv + j] * V[i * ldv + j];
    return 0;
}
/* the trace pass over the slab W (n rows, ncols columns = columns [col0, col0 + ncols) of the symmetric matrix): rows j >= column k */
int fvgp_hip_grad_trace_cols(fvgp_handle *h, int kernel_id, const double *x, int64_t n, int d, const double *theta, int ntheta,
                             const double *W, int64_t ldw, int64_t col0, int64_t ncols, const double *b, int64_t ldb, double *partial, double *grad) {
    (void)partial;
    if (!h) return -1;
    if (!x) return -3;
    if (!theta) return -6;
    if (!W) return -8;
    if (!grad) return -15;
    kdesc k;
    int rc = kdesc_from_theta(kernel_id, d, theta, ntheta, &k);
    if (rc) return rc;
    const int nk = k.iso ? 2 : d + 1;
    for (int i = 0; i < ntheta; ++i) grad[i] = 0.0;
    for (int64_t i = col0; i < n; ++i)
        for (int64_t j = col0; j <= i && j < col0 + ncols; ++j) {
            const double wt = (i == j ? 1.0 : 2.0) * (W[i * ldw + (j - col0)] - (b ? b[i * ldb] * b[j * ldb] : 0.0));
            double e2[FVGP_MAX_DIM], r2 = 0.0;
            for (int q = 0; q < d; ++q) { const double e = (x[i * d + q] - x[j * d + q]) * k.invl[q]; e2[q] = e * e; r2 += e2[q]; }
            const double r = sqrt(r2);
            double phi = radial(k.kind, r2), cf;
            if (k.kind == 0) cf = k.sig * phi;
            else if (k.kind == 1) cf = 3.0 * k.sig * exp(-sqrt(3.0) * r);
            else cf = (5.0 / 3.0) * k.sig * (1.0 + sqrt(5.0) * r) * exp(-sqrt(5.0) * r);
            grad[0] += wt * phi;
            for (int q = 0; q < d; ++q) grad[k.iso ? 1 : 1 + q] += wt * cf * e2[q] * k.invl[q];
        }
    for (int i = 0; i < nk; ++i) grad[i] *= 0.5;
    return 0;
}
int fvgp_hip_add_lower(fvgp_handle *h, double *A, int64_t n, int64_t lda, const double *B, int64_t ldb, double alpha) {
    if (!h) return -1;
    if (!A) return -2;
    if (n <= 0) return -3;
    if (lda < n) return -4;
    if (!B) return -5;
    if (ldb < n) return -6;
    for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j <= i; ++j) A[i * lda + j] += alpha * B[i * ldb + j];
    return 0;
}
int fvgp_hip_symmetrize(fvgp_handle *h, double *A, int64_t n, int64_t lda) {
    if (!h) return -1;
    if (!A) return -2;
    if (n <= 0) return -3;
    if (lda < n) return -4;
    for (int64_t i = 0; i < n; ++i) for (int64_t j = i + 1; j < n; ++j) A[i * lda + j] = A[j * lda + i];
    return 0;
}
