// Prices the parts of context_count_kernel: VARIANT 0 = product (ds_add_u32), 1 = plain LDS store instead of the
// atomic, 2 = no LDS traffic (the index is folded into a register).  hipcc -O3 --offload-arch=gfx950 -DVARIANT=n
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#if VARIANT == 1
#define DIG_CTX_BUMP(addr) (*(volatile __attribute__((address_space(3))) unsigned*)(uintptr_t)(addr) = 1u)
#elif VARIANT == 2
#define DIG_CTX_BUMP(addr) asm volatile("" ::"v"(addr))
#endif
#include "../../digdriver_amd/csrc/dig_context.hip"

int main()
{
    const int64_t nwin = 288000, window = 10000, nbases = nwin * window, n_words = nbases / 8 + 2;
    std::vector<uint32_t> h(n_words);
    uint64_t st = 88172645463325252ull;
    for (auto& w : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; w = (uint32_t)st & 0x33333333u; }
    h[0] = h[n_words - 1] = 0x44444444u;
    std::vector<int64_t> rs(nwin), re(nwin);
    for (int64_t i = 0; i < nwin; ++i) { rs[i] = i * window; re[i] = rs[i] + window; }
    std::vector<int32_t> rc(nwin, 0);
    std::vector<uint8_t> rm(nwin, 0);
    int64_t off = 0, len = nbases;
    uint32_t* dw; int64_t *doff, *dlen, *drs, *dre; int32_t *drc, *dout; uint8_t* drm;
    hipMalloc(&dw, n_words * 4); hipMalloc(&doff, 8); hipMalloc(&dlen, 8); hipMalloc(&drs, nwin * 8); hipMalloc(&dre, nwin * 8);
    hipMalloc(&drc, nwin * 4); hipMalloc(&drm, nwin); hipMalloc(&dout, nwin * 256);
    hipMemcpy(dw, h.data(), n_words * 4, hipMemcpyHostToDevice); hipMemcpy(doff, &off, 8, hipMemcpyHostToDevice);
    hipMemcpy(dlen, &len, 8, hipMemcpyHostToDevice); hipMemcpy(drs, rs.data(), nwin * 8, hipMemcpyHostToDevice);
    hipMemcpy(dre, re.data(), nwin * 8, hipMemcpyHostToDevice); hipMemcpy(drc, rc.data(), nwin * 4, hipMemcpyHostToDevice);
    hipMemcpy(drm, rm.data(), nwin, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) dig_count_contexts(dw, n_words, doff, dlen, 1, drc, drs, dre, drm, nwin, dout, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("variant %d: %.3f ms per pass, %.2f TB/s\n", VARIANT, ms / 5, nbases * 0.5 / (ms / 5 * 1e-3) / 1e12);
    return 0;
}
