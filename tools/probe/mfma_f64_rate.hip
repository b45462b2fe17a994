// Probe: sustained issue rate of v_mfma_f64_16x16x4_f64 (and 4x4x4) on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate16(double* out, int iters, double a, double b)
{
    double4_t acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = double4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
    }
    double s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void rate4(double* out, int iters, double a, double b)
{
    double acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[n], 0, 0, 0);
    }
    double s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void ratefma(double* out, int iters, double a, double b)
{
    double acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = fma(acc[n], a, b);
    }
    double s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double* out;
    (void)hipMalloc(&out, 8 * 1024 * 1024);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", cus, p.clockRate);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int which = 0; which < 3; ++which) {
            const dim3 grid(cus), block(256 * wps);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(rate16<6>, grid, block, 0, 0, out, iters, 1.0, 1e-9);
                if (which == 1) hipLaunchKernelGGL(rate4<8>, grid, block, 0, 0, out, iters, 1.0, 1e-9);
                if (which == 2) hipLaunchKernelGGL(ratefma<8>, grid, block, 0, 0, out, iters, 0.999, 1e-9);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double nacc = which == 0 ? 6 : 8;
            const double ops = (double)iters * nacc * wps;   // instructions per SIMD
            const double flop_per = which == 0 ? 2048.0 : which == 1 ? 512.0 : 128.0;
            printf("waves/SIMD %d %-10s %.3f ms  %.1f ns/instr/SIMD  %.2f TFLOP/s\n", wps,
                   which == 0 ? "mfma16x16x4" : which == 1 ? "mfma4x4x4" : "v_fma_f64", ms, ms * 1e6 / ops,
                   ops * flop_per * cus * 4 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
