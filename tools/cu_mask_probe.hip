// Which physical compute units does bit i of a hipExtStreamCreateWithCUMask mask enable on this device?
// Build: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void probe(uint32_t *out) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup alive a little so the dispatcher spreads the grid over every enabled CU
    long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const uint32_t *mask, const char *label) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", label); return; }
    const int nb = 4096;
    uint32_t *d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    for (int i = 0; i < nb; ++i) {
        uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(key).second) per_xcc[xcc]++;
    }
    printf("%-28s -> %3zu CUs; per XCC:", label, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    if (cus.size() <= 8) { printf("  [xcc/se/sh/cu:"); for (auto k : cus) printf(" %u/%u/%u/%u", k >> 12, (k >> 8) & 7, (k >> 4) & 1, k & 15); printf("]"); }
    printf("\n");
    hipFree(d); hipStreamDestroy(s);
}

int main() {
    uint32_t m[8];
    char label[64];
    for (int bit : {0, 1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 128, 255}) {
        for (int i = 0; i < 8; ++i) m[i] = 0;
        m[bit / 32] = 1u << (bit % 32);
        snprintf(label, sizeof label, "bit %d", bit);
        run(m, label);
    }
    for (int w = 0; w < 8; ++w) { for (int i = 0; i < 8; ++i) m[i] = 0; m[w] = 0xFFFFFFFFu; snprintf(label, sizeof label, "word %d full", w); run(m, label); }
    for (int i = 0; i < 8; ++i) m[i] = 0x80808080u; run(m, "every 8th bit (7 mod 8)");
    for (int i = 0; i < 8; ++i) m[i] = 0x7F7F7F7Fu; run(m, "all but every 8th bit");
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu; m[7] = 0; run(m, "all but word 7");
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu; run(m, "all");
    return 0;
}
