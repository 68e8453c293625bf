// Shader clock while the fp64 GEMM runs: a one-wave probe on its own stream samples the core-clock counter
// (s_memtime) against the constant 100 MHz counter (s_memrealtime) while libfvgp_hip's GEMM / the register-only
// MFMA loop / nothing runs on another stream.
// Build (repo root): hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe -Lfvgp_amd/csrc -lfvgp_hip -Wl,-rpath,$PWD/fvgp_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../include/fvgp_hip.h"

__global__ void probe(unsigned long long *out, long long wall_ticks) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    while ((long long)(wall_clock64() - w0) < wall_ticks) { __builtin_amdgcn_s_sleep(8); }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[0] = c1 - c0; out[1] = w1 - w0;
}

static double run_probe(hipStream_t s, unsigned long long *d, double ms) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, d, (long long)(ms * 1e5));
    hipStreamSynchronize(s);
    unsigned long long h[2];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    return (double)h[0] / (double)h[1] * 100.0;      // MHz (wall counter = 100 MHz)
}

int main() {
    hipStream_t ps, ws;
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&ws, hipStreamNonBlocking);
    unsigned long long *d; hipMalloc(&d, 16);
    fvgp_handle *h; fvgp_hip_create(&h, 0, ws);
    printf("idle:                 %.0f MHz\n", run_probe(ps, d, 20.0));
    // register-only MFMA loop on every CU
    double *out; hipMalloc(&out, 512 * 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 6; ++i) fvgp_hip_mfma_peak(h, out, 512, 200000);
        double mhz = run_probe(ps, d, 30.0);
        hipStreamSynchronize(ws);
        printf("mfma register loop:   %.0f MHz\n", mhz);
    }
    // the trailing-update GEMM shape: C (n x n, lower) -= A A^T, K = 1024
    const int64_t n = 32768, K = 1024;
    double *A, *C; hipMalloc(&A, n * K * 8); hipMalloc(&C, n * n * 8);
    hipMemset(A, 0, n * K * 8); hipMemset(C, 0, n * n * 8);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 12; ++i) fvgp_hip_gemm(h, 0, 0, 1, n, n, K, -1.0, A, K, A, K, 1.0, C, n);
        double mhz = run_probe(ps, d, 60.0);
        hipStreamSynchronize(ws);
        printf("fp64 SYRK K=1024:     %.0f MHz\n", mhz);
    }
    // time the same GEMM alone to pair the clock with a rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, ws);
    for (int i = 0; i < 6; ++i) fvgp_hip_gemm(h, 0, 0, 1, n, n, K, -1.0, A, K, A, K, 1.0, C, n);
    hipEventRecord(e1, ws); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tiles = (double)(n / 128) * (n / 128 + 1) / 2.0;
    printf("SYRK alone: %.2f ms per launch, %.1f TFLOP/s\n", ms / 6, tiles * 128 * 128 * 2 * K / (ms / 6 * 1e-3) / 1e12);
    fvgp_hip_destroy(h);
    return 0;
}
