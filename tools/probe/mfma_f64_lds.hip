// Probe: v_mfma_f64_16x16x4_f64 fed from LDS (B) and int->f64 converts (A), no global traffic (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: B from LDS + cvt A ; 1: B from LDS, A constant ; 2: B constant, cvt A
__global__ void k(double* out, int tiles, int seed)
{
    extern __shared__ double tab[];
    for (int idx = threadIdx.x; idx < 64 * 3 * 64; idx += blockDim.x) tab[idx] = 1e-3 * (idx & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double4_t acc[3] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
    int v = seed + lane;
    for (int t = 0; t < tiles; ++t) {
        int z;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        const double* tw = tab + z;
#pragma unroll 16
        for (int s = 0; s < 64; ++s) {
            double A = 1.5;
            if (MODE != 1) { A = (double)v; v = (v * 3 + 1) & 1023; }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const double b = (MODE == 2) ? 0.25 : tw[(s * 3 + nt) * 64 + lane];
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, b, acc[nt], 0, 0, 0);
            }
        }
    }
    double sum = 0;
    for (int n = 0; n < 3; ++n) sum += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
int main()
{
    double* out;
    (void)hipMalloc(&out, 8 * 1024 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles = 40;
    const size_t lds = 64 * 3 * 64 * 8;
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int waves = 4; waves <= 16; waves *= 2)
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), lds, 0, out, tiles, 7);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), lds, 0, out, tiles, 7);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), lds, 0, out, tiles, 7);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double mfmas_per_simd = (double)tiles * 192 * (waves / 4);
            printf("waves/CU %2d mode %d: %.3f ms, %.1f cycles per MFMA per SIMD (2.4 GHz)\n", waves, mode, ms,
                   ms * 1e-3 * 2.4e9 / mfmas_per_simd);
        }
    return 0;
}
