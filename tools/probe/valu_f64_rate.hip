// Probe: issue cost of the FP64 / integer VALU operations the statistics kernels are made of (developer tool).
// One wave per SIMD slot x {1, 2, 4}, 8 independent chains per lane, inline asm so that the compiler cannot fuse or
// strength-reduce.  Prints ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define DEF_KERNEL(NAME, ASM)                                                                  \
    __global__ void NAME(double* out, int iters, double a, double b)                            \
    {                                                                                           \
        double acc[CHAINS];                                                                     \
        for (int n = 0; n < CHAINS; ++n) acc[n] = threadIdx.x + n;                              \
        for (int it = 0; it < iters; ++it) {                                                    \
            _Pragma("unroll") for (int n = 0; n < CHAINS; ++n) asm volatile(ASM : "+v"(acc[n]) : "v"(a), "v"(b)); \
        }                                                                                       \
        double s = 0;                                                                           \
        for (int n = 0; n < CHAINS; ++n) s += acc[n];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
    }

DEF_KERNEL(k_fma, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL(k_add, "v_add_f64 %0, %0, %2")
DEF_KERNEL(k_mul, "v_mul_f64 %0, %0, %1")
DEF_KERNEL(k_mov, "v_mov_b64 %0, %1")
DEF_KERNEL(k_max, "v_max_f64 %0, %0, %1")
DEF_KERNEL(k_ldexp, "v_ldexp_f64 %0, %0, 1")
DEF_KERNEL(k_rcp, "v_rcp_f64 %0, %0")
DEF_KERNEL(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %1")
#define DEF_KERNEL32(NAME, ASM)                                                                \
    __global__ void NAME(double* out, int iters, double a, double b)                            \
    {                                                                                           \
        unsigned acc[CHAINS];                                                                   \
        const unsigned ua = (unsigned)a + threadIdx.x, ub = (unsigned)b + 3;                    \
        for (int n = 0; n < CHAINS; ++n) acc[n] = threadIdx.x + n;                              \
        for (int it = 0; it < iters; ++it) {                                                    \
            _Pragma("unroll") for (int n = 0; n < CHAINS; ++n) asm volatile(ASM : "+v"(acc[n]) : "v"(ua), "v"(ub)); \
        }                                                                                       \
        unsigned s = 0;                                                                         \
        for (int n = 0; n < CHAINS; ++n) s += acc[n];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
    }
__global__ void k_mad64(double* out, int iters, double a, double b)
{
    unsigned long long acc[CHAINS];
    const unsigned ua = (unsigned)a + threadIdx.x, ub = (unsigned)b + 3;
    for (int n = 0; n < CHAINS; ++n) acc[n] = threadIdx.x + n;
    for (int it = 0; it < iters; ++it) {
        _Pragma("unroll") for (int n = 0; n < CHAINS; ++n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[n]) : "v"(ua), "v"(ub) : "vcc");
    }
    unsigned long long s = 0;
    for (int n = 0; n < CHAINS; ++n) s += acc[n];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
}
DEF_KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL32(k_mov32, "v_mov_b32 %0, %1")
DEF_KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
__global__ void k_cmp(double* out, int iters, double a, double b)
{
    double acc[CHAINS];
    for (int n = 0; n < CHAINS; ++n) acc[n] = threadIdx.x + n;
    for (int it = 0; it < iters; ++it) {
        _Pragma("unroll") for (int n = 0; n < CHAINS; ++n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(acc[n]), "v"(a) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
DEF_KERNEL32(k_add32, "v_add_u32 %0, %0, %1")

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
    typedef void (*K)(double*, int, double, double);
    struct { const char* name; K k; } ks[] = {{"v_fma_f64", k_fma}, {"v_add_f64", k_add}, {"v_mul_f64", k_mul}, {"v_mov_b64", k_mov},
        {"v_max_f64", k_max}, {"v_ldexp_f64", k_ldexp}, {"v_rcp_f64", k_rcp}, {"v_lshl_add_u64", k_lshladd64},
        {"v_mad_u64_u32", k_mad64}, {"v_mul_lo_u32", k_mullo}, {"v_mov_b32", k_mov32}, {"v_cmp_lt_f64", k_cmp},
        {"v_add_u32", k_add32}, {"v_mul_u32_u24", k_mul24}};
    for (auto& kk : ks) {
        printf("%-16s", kk.name);
        for (int wps = 1; wps <= 4; wps *= 2) {
            const dim3 grid(cus), block(256 * wps);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(kk.k, grid, block, 0, 0, out, iters, 0.999, 1e-9);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)iters * CHAINS * wps;   // instructions per SIMD
            printf("  %d w/SIMD: %6.2f ns/instr", wps, ms * 1e6 / ops);
        }
        printf("\n");
    }
    return 0;
}
