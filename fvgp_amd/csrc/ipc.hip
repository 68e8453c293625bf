// Direct all-gather / all-reduce over peer mappings: the collectives of the row-sharded evaluation without a collective library's kernels.
//
// Stands in for the reference's direct worker-to-worker scatter / gather of covariance blocks (fvgp/gp_prior.py:301-322,
// gp2Scale_covariance.py:419-420).  Behind the same `fvgp_collectives` pointers as the RCCL binding (dist.hip), so
// dist_driver.h does not change.  Why it exists: beside the trailing update that fills the chip, every collective of a
// library queues its (register- and LDS-heavy) kernels behind the update's resident workgroups -- 0.34 ms of START latency per call
// measured, twice per panel (profiles/r04_rccl_beside_update.txt) -- and a ring moves a panel over ONE xGMI link at a time.  Here
// the payload moves by hipMemcpyAsync between peer mappings (the copy engines: one stream, one engine and one link per peer, all
// peers at once: the direct gather of SURVEY 8e), and the only kernels are one-wave flag writers and pollers with a handful of
// registers, which find a wave slot on a full chip.
//
// Protocol.  Every rank owns a WINDOW (two halves) allocated here with hipMalloc and exported with hipIpcGetMemHandle; the 64-byte
// handles travel over the bootstrap that also carries the RCCL id (torch.distributed's store); every rank maps every peer's window
// (hipIpcOpenMemHandle).  Flag words -- ready[r], done[r]: 64-bit sequence numbers, a 128-byte line each -- live in ONE POSIX
// shared-memory file every rank maps and registers with HIP (host memory: coherent for every GPU and every process).
// all_gather number s (half b = s & 1) on rank r, everything enqueued on the caller's stream and on per-peer copy streams:
//   1. poll done[q] >= s - 2 for every q     (the peers have pulled what this half held two calls ago)
//   2. copy the rank's chunk into window[b]; flag kernel ready[r] = s (system-scope store behind a system fence)
//   3. per peer q, on copy stream q (which first waits for the caller's stream): poll ready[q] >= s, then
//      hipMemcpyAsync(recv + q count, peer window q [b]); the caller's stream waits for every copy stream's event
//   4. flag kernel done[r] = s
// Every poll is bounded (60 s; FVGP_IPC_TIMEOUT_S): a rank that gives up raises the error word of the flag file, every later poll of any rank
// returns at once, and the host-side check behind the next synchronisation (fvgp_ipc_check: every entry that hands results to the
// host asks it AFTER waiting for the stream) reports status 2200 instead of hanging the GPU or returning what stale windows held.
// all_reduce = all_gather into a scratch + a sum in rank order (the same bits on every rank, whatever arrives first).
// Chunks: a call larger than half a window is cut into pieces, each a gather of its own.
#include "common.h"
#include <chrono>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int MAXR = 16;
constexpr int FL = 16;                 // 64-bit words between two flags (one 128-byte line each)
constexpr int F_READY = 0, F_DONE = MAXR, F_ERR = 2 * MAXR, F_WORDS = (2 * MAXR + 1) * FL;

struct IpcComm {
    int rank = 0, nranks = 1, device = 0;
    size_t half_doubles = 0;           // doubles per window half
    double *win = nullptr;             // this rank's window (two halves)
    double *peer[MAXR] = {};           // every rank's window as mapped here (peer[rank] == win)
    void *flags_host = nullptr; size_t flags_bytes = 0; int shm_fd = -1;
    unsigned long long *flags = nullptr;   // device pointer of the registered flag file
    unsigned long long seq = 0;
    unsigned long long timeout_ticks = 6000000000ull;     // 100 MHz ticks a poll waits before it gives up (FVGP_IPC_TIMEOUT_S, default 60 s)
    hipStream_t copy[MAXR] = {};
    hipEvent_t ev[MAXR] = {}, ev0 = nullptr;
    double *red = nullptr; size_t red_cap = 0;
};

__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane q < nranks waits until flags[(base + q) * FL] >= need (lanes with skip == q do not wait); bounded
__global__ void ipc_poll_kernel(unsigned long long *flags, int base, int nranks, int only, unsigned long long need, unsigned long long timeout_ticks) {
    const int q = threadIdx.x;
    if (q >= nranks || (only >= 0 && q != only)) return;
    const unsigned long long *p = flags + (size_t)(base + q) * FL;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while ((long long)(sys_load(p) - need) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (sys_load(flags + (size_t)F_ERR * FL) != 0ull) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { sys_store(flags + (size_t)F_ERR * FL, 1ull + (unsigned long long)q); return; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

__global__ void ipc_flag_kernel(unsigned long long *flags, int word, unsigned long long value) {
    __threadfence_system();
    sys_store(flags + (size_t)word * FL, value);
}

// out[i] = sum_q parts[q * stride + i], q ascending
__global__ void ipc_sum_kernel(double *out, const double *parts, int nranks, long stride, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        double s = parts[i];
        for (int q = 1; q < nranks; ++q) s += parts[(long)q * stride + i];
        out[i] = s;
    }
}

int ipc_gather_chunk(IpcComm *c, const double *send, double *recv, int64_t cnt, int64_t recv_stride, hipStream_t stream) {
    const unsigned long long s = ++c->seq;
    const int b = (int)(s & 1);
    const size_t bytes = (size_t)cnt * sizeof(double);
    double *mine = c->win + (size_t)b * c->half_doubles;
    if (s > 2) {
        hipLaunchKernelGGL(ipc_poll_kernel, dim3(1), dim3(64), 0, stream, c->flags, F_DONE, c->nranks, -1, s - 2, c->timeout_ticks);
    }
    HIPCHK(hipMemcpyAsync(mine, send, bytes, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(ipc_flag_kernel, dim3(1), dim3(1), 0, stream, c->flags, F_READY + c->rank, s);
    HIPCHK(hipMemcpyAsync(recv + (size_t)c->rank * recv_stride, send, bytes, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipEventRecord(c->ev0, stream));
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank) continue;
        HIPCHK(hipStreamWaitEvent(c->copy[q], c->ev0, 0));
        hipLaunchKernelGGL(ipc_poll_kernel, dim3(1), dim3(64), 0, c->copy[q], c->flags, F_READY, c->nranks, q, s, c->timeout_ticks);
        HIPCHK(hipMemcpyAsync(recv + (size_t)q * recv_stride, c->peer[q] + (size_t)b * c->half_doubles, bytes, hipMemcpyDeviceToDevice, c->copy[q]));
        HIPCHK(hipEventRecord(c->ev[q], c->copy[q]));
    }
    for (int q = 0; q < c->nranks; ++q)
        if (q != c->rank) HIPCHK(hipStreamWaitEvent(stream, c->ev[q], 0));
    hipLaunchKernelGGL(ipc_flag_kernel, dim3(1), dim3(1), 0, stream, c->flags, F_DONE + c->rank, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int ipc_check(IpcComm *c) {
    const volatile unsigned long long *err = reinterpret_cast<const volatile unsigned long long *>(c->flags_host) + (size_t)F_ERR * FL;
    if (*err != 0ull) {
        fvgp_set_error("ipc collectives: a rank gave up waiting for a peer's flag (rank index + 1 = " + std::to_string(*err) +
                       "); what the collectives since then delivered is stale and the communicator stays unusable: build a new one");
        return 2200;
    }
    return 0;
}

int ipc_all_gather(void *ctx, const double *send, double *recv, int64_t count, void *stream) {
    IpcComm *c = static_cast<IpcComm *>(ctx);
    int rc = ipc_check(c); if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    for (int64_t off = 0; off < count; off += (int64_t)c->half_doubles) {
        const int64_t cnt = count - off < (int64_t)c->half_doubles ? count - off : (int64_t)c->half_doubles;
        rc = ipc_gather_chunk(c, send + off, recv + off, cnt, count, (hipStream_t)stream); if (rc) return rc;
    }
    return 0;
}

int ipc_all_reduce(void *ctx, double *buf, int64_t count, void *stream) {
    IpcComm *c = static_cast<IpcComm *>(ctx);
    int rc = ipc_check(c); if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    const int64_t chunk = (int64_t)c->half_doubles;
    const size_t need = (size_t)c->nranks * (size_t)(count < chunk ? count : chunk);
    if (need > c->red_cap) {
        // (grows on the first calls only; the free synchronises the device, like every scratch of the handle)
        if (c->red) HIPCHK(hipFree(c->red));
        c->red = nullptr; c->red_cap = 0;
        HIPCHK(hipMalloc((void **)&c->red, need * sizeof(double)));
        c->red_cap = need;
    }
    for (int64_t off = 0; off < count; off += chunk) {
        const int64_t cnt = count - off < chunk ? count - off : chunk;
        rc = ipc_gather_chunk(c, buf + off, c->red, cnt, cnt, (hipStream_t)stream); if (rc) return rc;
        long blocks = (cnt + 255) / 256; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(ipc_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, buf + off, c->red, c->nranks, (long)cnt, (long)cnt);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

}  // namespace

// The error word after the host has waited for the stream: a poll that gives up lets the copies behind it run on stale window
// contents, so every entry that hands results to the host (fvgp_read_back, fvgp_hip_sync, the _dist entries) asks here AFTER its
// synchronisation and fails with 2200 instead of returning those results.
int fvgp_ipc_check(fvgp_handle *h) {
    IpcComm *c = static_cast<IpcComm *>(h->ipc_comm);
    if (!c || !c->flags_host) return 0;
    return ipc_check(c);
}

void fvgp_ipc_destroy(fvgp_handle *h) {
    IpcComm *c = static_cast<IpcComm *>(h->ipc_comm);
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->flags_host && c->seq > 0) {
        // a slower peer may still be pulling gather number seq (or seq - 1) out of this rank's window: done[q] is only awaited two
        // calls later.  Wait (bounded, on the host) until every peer has pulled the last gather before the window is freed.
        const volatile unsigned long long *fl = reinterpret_cast<const volatile unsigned long long *>(c->flags_host);
        const auto t0 = std::chrono::steady_clock::now();
        for (int q = 0; q < c->nranks; ++q) {
            if (q == c->rank) continue;
            while ((long long)(fl[(size_t)(F_DONE + q) * FL] - c->seq) < 0 && fl[(size_t)F_ERR * FL] == 0ull &&
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0)
                usleep(200);
        }
    }
    for (int q = 0; q < c->nranks; ++q) {
        if (c->copy[q]) (void)hipStreamDestroy(c->copy[q]);
        if (c->ev[q]) (void)hipEventDestroy(c->ev[q]);
        if (q != c->rank && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->flags_host) { (void)hipHostUnregister(c->flags_host); munmap(c->flags_host, c->flags_bytes); }
    if (c->shm_fd >= 0) close(c->shm_fd);
    if (c->red) (void)hipFree(c->red);
    if (c->win) (void)hipFree(c->win);
    delete c;
    h->ipc_comm = nullptr;
}

extern "C" {

int fvgp_hip_ipc_window(fvgp_handle *h, int64_t window_bytes, void *out_handle64_host) {
    if (!h) return -1;
    if (window_bytes < 2 * 1024 || window_bytes % 32) { fvgp_set_error("ipc_window: the window needs a multiple of 32 bytes, at least 2 KB"); return -2; }
    if (!out_handle64_host) return -3;
    HIPCHK(hipSetDevice(h->device));
    (void)fvgp_hip_comm_destroy(h);
    IpcComm *c = new IpcComm();
    c->device = h->device;
    c->half_doubles = (size_t)window_bytes / 2 / sizeof(double);
    if (const char *t = getenv("FVGP_IPC_TIMEOUT_S")) { const double sec = atof(t); if (sec >= 1.0 && sec <= 3600.0) c->timeout_ticks = (unsigned long long)(sec * 1e8); }
    h->ipc_comm = c;                 // owned by the handle from here on: an error below is cleaned up by fvgp_ipc_destroy
    hipIpcMemHandle_t hd;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipError_t e = hipMalloc((void **)&c->win, (size_t)window_bytes);
    if (e == hipSuccess) e = hipMemset(c->win, 0, (size_t)window_bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hd, c->win);
    if (e != hipSuccess) { fvgp_ipc_destroy(h); return fvgp_hip_fail(e, "ipc_window: hipMalloc / hipIpcGetMemHandle", __LINE__); }
    memcpy(out_handle64_host, &hd, sizeof(hd));
    return 0;
}

static int comm_init_ipc_body(fvgp_handle *h, const void *all_handles64_host, const char *shm_name, int rank, int nranks) {
    if (!h) return -1;
    if (!all_handles64_host) return -2;
    if (!shm_name) return -3;
    if (nranks < 1 || nranks > MAXR) { fvgp_set_error("comm_init_ipc: 1 <= nranks <= 16"); return -5; }
    if (rank < 0 || rank >= nranks) return -4;
    IpcComm *c = static_cast<IpcComm *>(h->ipc_comm);
    if (!c) { fvgp_set_error("comm_init_ipc: call fvgp_hip_ipc_window first"); return -1; }
    HIPCHK(hipSetDevice(h->device));
    c->rank = rank; c->nranks = nranks;
    // the flag file: created (and zeroed: ftruncate) by whoever comes first, mapped by everybody
    c->flags_bytes = (size_t)F_WORDS * sizeof(unsigned long long);
    c->shm_fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
    if (c->shm_fd < 0) { fvgp_set_error(std::string("comm_init_ipc: shm_open failed for ") + shm_name); return 2201; }
    if (ftruncate(c->shm_fd, (off_t)c->flags_bytes) != 0) { fvgp_set_error("comm_init_ipc: ftruncate failed"); return 2201; }
    c->flags_host = mmap(nullptr, c->flags_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    if (c->flags_host == MAP_FAILED) { c->flags_host = nullptr; fvgp_set_error("comm_init_ipc: mmap failed"); return 2201; }
    HIPCHK(hipHostRegister(c->flags_host, c->flags_bytes, hipHostRegisterMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&c->flags, c->flags_host, 0));
    const hipIpcMemHandle_t *hd = static_cast<const hipIpcMemHandle_t *>(all_handles64_host);
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) { c->peer[q] = c->win; continue; }
        void *p = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, hd[q], hipIpcMemLazyEnablePeerAccess));
        c->peer[q] = static_cast<double *>(p);
        HIPCHK(hipStreamCreateWithFlags(&c->copy[q], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev[q], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev0, hipEventDisableTiming));
    h->coll = fvgp_collectives{c, ipc_all_gather, ipc_all_reduce};
    h->coll_rank = rank; h->coll_nranks = nranks;
    return 0;
}

int fvgp_hip_comm_init_ipc(fvgp_handle *h, const void *all_handles64_host, const char *shm_name, int rank, int nranks) {
    const int rc = comm_init_ipc_body(h, all_handles64_host, shm_name, rank, nranks);
    if (rc != 0 && h && h->ipc_comm) fvgp_ipc_destroy(h);        // the window, the flag file's mapping and descriptor, the peers opened so far
    return rc;
}

}  // extern "C"
