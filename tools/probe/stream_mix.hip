// Probe: which property of the statistics kernel's memory pattern costs the time?  A persistent grid with the same tile
// walk streams KI f64 planes + KJ i32 planes in and KO f64 planes + KP i32 planes out; plane strides of inputs and
// outputs are chosen separately (aligned = multiple of 64 elements, or the raw odd E*C).
//   hipcc -O3 --offload-arch=gfx950 stream_mix.hip -o stream_mix && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KI, int KJ, int KO, int KP>
__global__ __launch_bounds__(1024) void mix(const double* __restrict__ in, const int* __restrict__ iin, double* __restrict__ out,
                                            int* __restrict__ iout, long n, long sin, long sout)
{
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long n_tiles = n / 64;
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        const long tile = ((long)t * 8 + (blockIdx.x & 7)) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (tile >= n_tiles) break;
        const long i = tile * 64 + lane;
        double v = 0.0;
        int w = 0;
#pragma unroll
        for (int j = 0; j < KI; ++j) v += in[(long)j * sin + i];
#pragma unroll
        for (int j = 0; j < KJ; ++j) w += iin[(long)j * sin + i];
#pragma unroll
        for (int j = 0; j < KO; ++j) __builtin_nontemporal_store(v + j + w, &out[(long)j * sout + i]);
#pragma unroll
        for (int j = 0; j < KP; ++j) __builtin_nontemporal_store(w + j, &iout[(long)j * sout + i]);
    }
}

// the statistics kernel's own output set: MU, SIGMA (f64) and R_OBS, FLAG (i32) in aligned arrays of their own, seven planes at
// stride `sout`
__global__ __launch_bounds__(1024) void mix_real(const double* __restrict__ in, const int* __restrict__ iin, double* __restrict__ out,
                                                 int* __restrict__ iout, long n, long sin, long sout, long pad)
{
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long n_tiles = n / 64;
    double* planes = out + 2 * pad;
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        const long tile = ((long)t * 8 + (blockIdx.x & 7)) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (tile >= n_tiles) break;
        const long i = tile * 64 + lane;
        double v = in[i] + in[sin + i];
        int w = iin[i] + iin[sin + i] + iin[2 * sin + i];
        __builtin_nontemporal_store(v, &out[i]);
        __builtin_nontemporal_store(v + 1, &out[pad + i]);
        __builtin_nontemporal_store(w, &iout[i]);
        __builtin_nontemporal_store(w + 1, &iout[pad + i]);
#pragma unroll
        for (int j = 0; j < 7; ++j) __builtin_nontemporal_store(v + j + w, &planes[(long)j * sout + i]);
    }
}

template <int KI, int KJ, int KO, int KP>
static void run(const char* what, long n, long sin, long sout, double* in, int* iin, double* out, int* iout, int grid)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((mix<KI, KJ, KO, KP>), dim3(grid), dim3(1024), 0, 0, in, iin, out, iout, n, sin, sout);
    (void)hipEventRecord(a, 0);
    const int K = 100;
    for (int w = 0; w < K; ++w) hipLaunchKernelGGL((mix<KI, KJ, KO, KP>), dim3(grid), dim3(1024), 0, 0, in, iin, out, iout, n, sin, sout);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (8.0 * (KI + KO) + 4.0 * (KJ + KP)) * n;
    printf("%-58s in %d f64 + %d i32, out %d f64 + %d i32: %6.1f us  %5.0f GB/s  (%.0f MB)\n", what, KI, KJ, KO, KP, ms / K * 1e3,
           bytes / (ms / K * 1e-3) / 1e9, bytes / 1e6);
}

int main()
{
    const long n = 120091L * 37;                         // 4 443 367 (odd)
    const long pad = (n + 63) / 64 * 64;
    double *in, *out;
    int *iin, *iout;
    (void)hipMalloc(&in, 8 * 16 * pad);
    (void)hipMalloc(&out, 8 * 16 * pad);
    (void)hipMalloc(&iin, 4 * 16 * pad);
    (void)hipMalloc(&iout, 4 * 16 * pad);
    (void)hipMemset(in, 0, 8 * 16 * pad);
    (void)hipMemset(iin, 0, 4 * 16 * pad);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<5, 0, 13, 0>("aligned in, aligned out", n, pad, pad, in, iin, out, iout, grid);
        run<5, 0, 13, 0>("aligned in, odd out stride", n, pad, n, in, iin, out, iout, grid);
        run<5, 0, 13, 0>("odd in stride, aligned out", n, n, pad, in, iin, out, iout, grid);
        run<5, 0, 13, 0>("odd in, odd out", n, n, n, in, iin, out, iout, grid);
        run<2, 3, 9, 2>("statistics mix, aligned", n, pad, pad, in, iin, out, iout, grid);
        run<2, 3, 9, 2>("statistics mix, odd out stride", n, pad, n, in, iin, out, iout, grid);
        run<2, 3, 9, 0>("statistics mix without the i32 outputs, aligned", n, pad, pad, in, iin, out, iout, grid);
        run<2, 3, 7, 0>("seven planes only, aligned", n, pad, pad, in, iin, out, iout, grid);
        run<2, 3, 0, 0>("reads only", n, pad, pad, in, iin, out, iout, grid);
        run<0, 0, 9, 2>("writes only", n, pad, pad, in, iin, out, iout, grid);
        for (int odd = 0; odd < 2; ++odd) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            const long so = odd ? n : pad;
            for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(mix_real, dim3(grid), dim3(1024), 0, 0, in, iin, out, iout, n, pad, so, pad);
            (void)hipEventRecord(a, 0);
            for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(mix_real, dim3(grid), dim3(1024), 0, 0, in, iin, out, iout, n, pad, so, pad);
            (void)hipEventRecord(b, 0);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("the kernel's own set: MU SIGMA R_OBS FLAG aligned, planes at %s stride: %6.1f us\n", odd ? "odd E*C   " : "padded    ", ms / 100 * 1e3);
        }
    }
    return 0;
}
