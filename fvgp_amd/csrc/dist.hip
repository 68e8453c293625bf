// Row-sharded evaluation behind the C ABI: the HIP backend of dist_driver.h, the RCCL binding and the collective entries.
//
// Stands in for the reference's distribution switch GP(..., gp2Scale=True, dask_client=...) (fvgp/gp.py:419-439) and its
// scheduler (gp_prior.py:324-347, gp2Scale_covariance.py:313-431): one process per GPU, the panel chain (gather of the
// diagonal block, factorisation of the tall panel, all-gather of the panel factor) on the handle's high-priority stream,
// the trailing updates on its main stream, RCCL called directly on the chain stream.  librccl is opened at run time
// (dlopen) by fvgp_hip_comm_init only: a single-GPU process never loads it.
#include "common.h"
#include <chrono>
#include "dist_driver.h"
#include <dlfcn.h>
#include <math.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {

// A[i][g(i)] += v[g] on the rank's data rows, = 1 on the padding rows of the last block (g = global index of local row i)
__global__ void dist_diag_kernel(double *A, long ld, long nrows, int P, int p, long n, long np, const double *v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const long g = ((i >> 7) * P + p) * 128 + (i & 127);
    if (g < n) A[i * ld + g] += v[g];
    else if (g < np) A[i * ld + g] = 1.0;
}

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *);
    ncclResult_t (*GetVersion)(int *);
    void *lib = nullptr;
};
Rccl g_rccl;

int rccl_open() {
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fvgp_set_error(std::string("cannot open librccl: ") + dlerror()); return 2001; }
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(lib, name); \
        if (!g_rccl.field) { fvgp_set_error(std::string("librccl lacks ") + name); dlclose(lib); return 2002; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
    SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank"); SYM(CommCuDevice, "ncclCommCuDevice"); SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_rccl.lib = lib;
    return 0;
}

int rccl_fail(ncclResult_t r, const char *what) {
    fvgp_set_error(std::string("RCCL error '") + g_rccl.GetErrorString(r) + "' in " + what);
    return 2100 + (int)r;
}

int rccl_all_gather(void *ctx, const double *send, double *recv, int64_t count, void *stream) {
    const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)count, ncclDouble, (ncclComm_t)ctx, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(r, "ncclAllGather");
}

int rccl_all_reduce(void *ctx, double *buf, int64_t count, void *stream) {
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(r, "ncclAllReduce");
}

// a collective on `stream`, timed with events when the handle profiles
int timed_collective(fvgp_handle *h, int kind, double bytes, hipStream_t stream, const double *send, double *buf, int64_t count) {
    if (kind == 0 ? !h->coll.all_gather : !h->coll.all_reduce_sum) {
        if (h->coll_nranks == 1) {                 // no communicator on a single rank: the collective is a copy / nothing
            if (kind == 0 && buf != send) HIPCHK(hipMemcpyAsync(buf, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, stream));
            return 0;
        }
        fvgp_set_error("no collectives bound to this handle: call fvgp_hip_comm_init first"); return 2003;
    }
    fvgp_handle::CollRec rec{kind, bytes, nullptr, nullptr};
    const bool timed = h->profile && h->coll_rec.size() < 65536;      // a caller that never polls fvgp_hip_comm_profile stops collecting events there
    if (timed) {
        for (hipEvent_t *e : {&rec.e0, &rec.e1}) {
            if (!h->coll_ev_pool.empty()) { *e = h->coll_ev_pool.back(); h->coll_ev_pool.pop_back(); }
            else HIPCHK(hipEventCreate(e));
        }
        HIPCHK(hipEventRecord(rec.e0, stream));
    }
    const int rc = kind == 0 ? h->coll.all_gather(h->coll.ctx, send, buf, count, stream) : h->coll.all_reduce_sum(h->coll.ctx, buf, count, stream);
    if (rc) return rc;
    if (timed) { HIPCHK(hipEventRecord(rec.e1, stream)); h->coll_rec.push_back(rec); }
    return 0;
}

__global__ void dist_identity_kernel(double *p, long ld, long n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) { const long i = e / n, j = e - i * n; p[i * ld + j] = i == j ? 1.0 : 0.0; }
}

// the operations dist_driver.h sequences, on the handle's two streams
struct HipBackend {
    fvgp_handle *h;
    hipStream_t mainS, chainS;
    int use_chain(bool c) { h->stream = c ? chainS : mainS; return 0; }
    int fork() {          // the chain stream waits for everything enqueued on the main stream so far
        HIPCHK(hipEventRecord(h->ev_cols, mainS));
        HIPCHK(hipStreamWaitEvent(chainS, h->ev_cols, 0));
        return 0;
    }
    int join() {          // the main stream waits for everything enqueued on the chain stream so far
        HIPCHK(hipEventRecord(h->ev_panel, chainS));
        HIPCHK(hipStreamWaitEvent(mainS, h->ev_panel, 0));
        return 0;
    }
    int zero(double *p, int64_t count) {
        HIPCHK(hipMemsetAsync(p, 0, (size_t)count * sizeof(double), h->stream));
        return 0;
    }
    int zero2d(double *p, int64_t ld, int64_t rows, int64_t cols) {
        HIPCHK(hipMemset2DAsync(p, (size_t)ld * sizeof(double), 0, (size_t)cols * sizeof(double), (size_t)rows, h->stream));
        return 0;
    }
    int identity2d(double *p, int64_t ld, int64_t n) {          // the n x n block at p (leading dimension ld) <- I
        hipLaunchKernelGGL(dist_identity_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, h->stream, p, (long)ld, (long)n);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int identity(double *p, int64_t n) { return identity2d(p, n, n); }
    int to_device(double *dst, const double *src_host, int64_t count) {
        HIPCHK(hipMemcpyAsync(dst, src_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return 0;
    }
    int to_host(double *dst_host, const double *src, int64_t count) {
        HIPCHK(hipMemcpyAsync(dst_host, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return fvgp_ipc_check(h);
    }
    int copy2d(double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols) {
        if (rows <= 0 || cols <= 0) return 0;
        if (ldd == cols && lds == cols) {         // contiguous on both sides: one linear copy (the rectangle path costs ~40 us per call)
            HIPCHK(hipMemcpyAsync(dst, src, (size_t)rows * cols * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            return 0;
        }
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(double), src, (size_t)lds * sizeof(double), (size_t)cols * sizeof(double),
                                (size_t)rows, hipMemcpyDeviceToDevice, h->stream));
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
        hipLaunchKernelGGL(dist_diag_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, h->stream, A, (long)ld, (long)nrows, P, p,
                           (long)n, (long)np, v);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int panel_potrf(double *T, int64_t w, int64_t rows, int64_t ldt, int64_t n_valid, int *info_dev, double *logdet_dev) {
        return fvgp_hip_panel_potrf_dev(h, T, w, rows, ldt, n_valid, info_dev, logdet_dev);
    }
    int syrk(int64_t M, int64_t N, int64_t K, const double *A, int64_t lda, const double *Bm, int64_t ldb, double *C, int64_t ldc,
             int scale, int off, int b_ranks, int b_blocks, int b_off) {
        return fvgp_hip_syrk_rowshard(h, M, N, K, A, lda, Bm, ldb, C, ldc, scale, off, b_ranks, b_blocks, b_off);
    }
    int all_gather(const double *send, double *recv, int64_t count, double bytes) {
        return timed_collective(h, 0, bytes, h->stream, send, recv, count);
    }
};

}  // namespace

extern "C" {

int fvgp_hip_comm_unique_id(void *out128_host) {
    if (!out128_host) return -1;
    int rc = rccl_open(); if (rc) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128_host, &id, sizeof(id));
    return 0;
}

int fvgp_hip_comm_destroy(fvgp_handle *h) {
    if (!h) return -1;
    if (h->side) (void)hipStreamSynchronize(h->side);         // the chain stream issues the collectives: it must be done with the communicator
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->rccl_comm) { (void)g_rccl.CommDestroy((ncclComm_t)h->rccl_comm); h->rccl_comm = nullptr; }
    fvgp_ipc_destroy(h);
    h->coll = fvgp_collectives{nullptr, nullptr, nullptr};
    h->coll_rank = 0; h->coll_nranks = 1;
    for (auto &r : h->coll_rec) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    h->coll_rec.clear();
    for (auto e : h->coll_ev_pool) (void)hipEventDestroy(e);
    h->coll_ev_pool.clear();
    return 0;
}

int fvgp_hip_comm_init(fvgp_handle *h, const void *unique_id128_host, int rank, int nranks) {
    if (!h) return -1;
    if (!unique_id128_host) return -2;
    if (nranks < 1) return -4;
    if (rank < 0 || rank >= nranks) return -3;
    HIPCHK(hipSetDevice(h->device));
    int rc = rccl_open(); if (rc) return rc;
    (void)fvgp_hip_comm_destroy(h);
    ncclUniqueId id;
    memcpy(&id, unique_id128_host, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    h->rccl_comm = comm;
    h->coll = fvgp_collectives{comm, rccl_all_gather, rccl_all_reduce};
    h->coll_rank = rank; h->coll_nranks = nranks;
    return 0;
}

int fvgp_hip_comm_init_callbacks(fvgp_handle *h, const fvgp_collectives *cb, int rank, int nranks) {
    if (!h) return -1;
    if (!cb || !cb->all_gather || !cb->all_reduce_sum) return -2;
    if (nranks < 1) return -4;
    if (rank < 0 || rank >= nranks) return -3;
    (void)fvgp_hip_comm_destroy(h);
    h->coll = *cb;
    h->coll_rank = rank; h->coll_nranks = nranks;
    return 0;
}

int fvgp_hip_comm_info(fvgp_handle *h, int64_t *out8) {
    if (!h) return -1;
    if (!out8) return -2;
    for (int i = 0; i < 8; ++i) out8[i] = -1;
    out8[0] = h->rccl_comm ? 1 : h->ipc_comm && h->coll.all_gather ? 2 : h->coll.all_gather ? 3 : 0;
    out8[1] = h->coll_nranks; out8[2] = h->coll_rank;
    if (h->rccl_comm) {
        // what the communicator itself says -- not what this library was told
        int v = -1;
        ncclResult_t r = g_rccl.CommCount((ncclComm_t)h->rccl_comm, &v); if (r != ncclSuccess) return rccl_fail(r, "ncclCommCount");
        out8[3] = v;
        r = g_rccl.CommUserRank((ncclComm_t)h->rccl_comm, &v); if (r != ncclSuccess) return rccl_fail(r, "ncclCommUserRank");
        out8[4] = v;
        r = g_rccl.CommCuDevice((ncclComm_t)h->rccl_comm, &v); if (r != ncclSuccess) return rccl_fail(r, "ncclCommCuDevice");
        out8[5] = v;
        r = g_rccl.GetVersion(&v); if (r != ncclSuccess) return rccl_fail(r, "ncclGetVersion");
        out8[6] = v;
    }
    return 0;
}

int fvgp_hip_comm_check(fvgp_handle *h) {
    if (!h) return -1;
    return fvgp_ipc_check(h);
}

int fvgp_hip_all_reduce(fvgp_handle *h, double *buf, int64_t count) {
    if (!h) return -1;
    if (!buf) return -2;
    if (count <= 0) return -3;
    HIPCHK(hipSetDevice(h->device));
    return timed_collective(h, 1, 16.0 * (h->coll_nranks - 1) / h->coll_nranks * (double)count, h->stream, buf, buf, count);
}

int fvgp_hip_all_gather(fvgp_handle *h, const double *send, double *recv, int64_t count_per_rank) {
    if (!h) return -1;
    if (!send) return -2;
    if (!recv) return -3;
    if (count_per_rank <= 0) return -4;
    HIPCHK(hipSetDevice(h->device));
    return timed_collective(h, 0, 8.0 * (h->coll_nranks - 1) * (double)count_per_rank, h->stream, send, recv, count_per_rank);
}

int fvgp_hip_comm_profile(fvgp_handle *h, double *out) {
    if (!h) return -1;
    if (!out) return -2;
    HIPCHK(hipSetDevice(h->device));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 6; ++i) out[i] = 0.0;
    for (auto &r : h->coll_rec) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        out[r.kind] += 1.0; out[2 + r.kind] += r.bytes; out[4 + r.kind] += ms;
        h->coll_ev_pool.push_back(r.e0); h->coll_ev_pool.push_back(r.e1);
    }
    h->coll_rec.clear();
    return 0;
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
    if (d->panel < FVGP_TILE || d->panel % FVGP_TILE) { fvgp_set_error("loglik_dist: the panel width must be a positive multiple of 128"); return -2; }
    if (d->nranks != h->coll_nranks || d->rank != h->coll_rank) {
        fvgp_set_error("loglik_dist: rank / nranks of the descriptor differ from the handle's communicator (fvgp_hip_comm_init)"); return -2;
    }
    if (!d->x_all || !d->vdiag || !d->zt || !d->A || !d->info_dev || !d->logdet_dev) return -2;
    const bool general = d->nranks > 1 || d->force_general;
    if (general && (!d->T[0] || !d->T[1] || !d->recv[0] || !d->recv[1] || !d->Dfac || !d->gather)) return -2;
    if (!theta) return -3;
    if (!out_host) return -5;
    HIPCHK(hipSetDevice(h->device));
    int rc = fvgp_ensure_side(h); if (rc) return rc;
    const fvgp_dist::Geom g = fvgp_dist::geometry(*d);
    // everything that can be refused is refused BEFORE the first launch: a rank that returned early from the middle of the
    // factorisation would leave its peers blocked in an all-gather
    if (g.npan > 2048) { fvgp_set_error("loglik_dist: more than 2048 panels"); return -2; }
    if (d->nranks > 1 && (!h->coll.all_gather || !h->coll.all_reduce_sum)) {
        fvgp_set_error("no collectives bound to this handle: call fvgp_hip_comm_init first"); return 2003;
    }
    HipBackend b{h, h->stream, h->side};
    HIPCHK(hipMemsetAsync(d->info_dev, 0, (size_t)g.npan * sizeof(int), b.mainS));
    HIPCHK(hipMemsetAsync(d->logdet_dev, 0, (size_t)g.npan * sizeof(double), b.mainS));
    const auto host_t0 = std::chrono::steady_clock::now();
    rc = fvgp_dist::evaluate(b, *d, theta, ntheta);
    h->prof_host_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    h->stream = b.mainS;
    if (rc) return rc;
    // scalars: |z|^2 over the appended rows, the per-panel log-dets, the per-panel info -- one host round trip
    rc = launch_rowsumsq(h, d->A, g.ld, g.zrow, d->ncol, g.n, h->red + 1); if (rc) return rc;
    rc = launch_sum(h, d->logdet_dev, g.npan, h->red); if (rc) return rc;
    int *hinfo = reinterpret_cast<int *>(h->hpin + 1024);
    HIPCHK(hipMemcpyAsync(hinfo, d->info_dev, (size_t)g.npan * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    double r[2];
    rc = fvgp_read_back(h, h->red, r, 2); if (rc) return rc;
    int info = 0;
    for (int J = 0; J < g.npan && !info; ++J) if (hinfo[J] != 0) info = (int)(g.bnd(J) + hinfo[J]);
    if (info_host) *info_host = info;
    if (info != 0) { out_host[0] = out_host[1] = out_host[2] = NAN; return 0; }
    const double logdet = r[0], quad = r[1] / (double)d->ncol;
    out_host[0] = -0.5 * (quad + logdet + (double)g.n * log(2.0 * M_PI));
    out_host[1] = logdet;
    out_host[2] = quad;
    return 0;
}

static int dist_after_factor_check(fvgp_handle *h, const fvgp_dist_desc *d) {
    if (!h) return -1;
    if (!d) return -2;
    if (d->n <= 0 || d->d < 1 || d->d > FVGP_MAX_DIM || d->panel < FVGP_TILE || d->panel % FVGP_TILE || !d->A || !d->x_all) return -2;
    if (d->nranks != h->coll_nranks || d->rank != h->coll_rank) {
        fvgp_set_error("dist: rank / nranks of the descriptor differ from the handle's communicator (fvgp_hip_comm_init)"); return -2;
    }
    if ((d->nranks > 1 || d->force_general) && !d->Dfac) return -2;
    if (d->nranks > 1 && (!h->coll.all_gather || !h->coll.all_reduce_sum)) {
        fvgp_set_error("no collectives bound to this handle: call fvgp_hip_comm_init first"); return 2003;
    }
    if (!d->keep_factor) { fvgp_set_error("dist: the last evaluation did not keep its factor (keep_factor = 0)"); return -2; }
    // the handle scratch the sweeps below draw on (split-K partials of fvgp_hip_gemm: at most 512 tiles; the block sweep of
    // fvgp_hip_trsm_lower: as much again) is sized ONCE here: growing it later means hipFree + hipMalloc, a device-wide
    // synchronisation in the middle of a multi-rank sequence of collectives
    HIPCHK(hipSetDevice(h->device));
    return ensure_scratch(h, 2 * 512 * (int64_t)LEAF_DOUBLES / 8 + 1024);
}

int64_t fvgp_hip_dist_scratch(const fvgp_dist_desc *d, int what, int64_t npred, int64_t slab) {
    if (!d || what < 0 || what > 2 || npred < 0 || (what == 2 && (slab < FVGP_TILE || slab % FVGP_TILE))) return -1;
    return fvgp_dist::scratch_doubles(*d, what, npred, slab);
}

int fvgp_hip_solve_dist(fvgp_handle *h, const fvgp_dist_desc *d, double *alpha_out, double *ws) {
    int rc = dist_after_factor_check(h, d); if (rc) return rc;
    if (!alpha_out) return -3;
    if (!ws) return -4;
    HIPCHK(hipSetDevice(h->device));
    HipBackend b{h, h->stream, h->side};
    return fvgp_dist::solve_backward(b, h, *d, alpha_out, ws);
}

int fvgp_hip_posterior_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta, int ntheta, const double *xpred, int64_t npred,
                            const double *k_pre, const double *kk_pre, const double *alpha, double *mean_out, double *S_out, double *ws) {
    int rc = dist_after_factor_check(h, d); if (rc) return rc;
    if (!theta) return -3;
    if (npred <= 0) return -6;
    if (!xpred && !(k_pre && (kk_pre || !S_out || d->rank != 0))) return -5;
    if (!alpha) return -9;
    if (!mean_out) return -10;
    if (!ws) return -12;
    HIPCHK(hipSetDevice(h->device));
    HipBackend b{h, h->stream, h->side};
    return fvgp_dist::posterior(b, h, *d, theta, ntheta, xpred, npred, k_pre, kk_pre, alpha, mean_out, S_out, ws);
}

int fvgp_hip_grad_dist(fvgp_handle *h, const fvgp_dist_desc *d, const double *theta, int ntheta, const double *alpha, int component,
                       int64_t slab, double *grad_host, double *diag_out, double *ws) {
    int rc = dist_after_factor_check(h, d); if (rc) return rc;
    if (!theta || ntheta < 1 || ntheta > FVGP_MAX_DIM + 1) return -3;
    if (!alpha) return -5;
    if (component < 0 || component >= FVGP_TILE) return -6;
    if (slab < FVGP_TILE || slab % FVGP_TILE) return -7;
    if (!grad_host) return -8;
    if (!ws) return -10;
    HIPCHK(hipSetDevice(h->device));
    HipBackend b{h, h->stream, h->side};
    return fvgp_dist::gradient(b, h, *d, theta, ntheta, alpha, component, slab, grad_host, diag_out, ws);
}

}  // extern "C"
