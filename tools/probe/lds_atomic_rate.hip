// Probe (developer tool): what an LDS atomic add without return costs on gfx950, by the number of waves of the CU issuing them
// and by the address pattern of the histogram walk (dig_tiles.hip: row = context, 136 dwords apart; column = lane / 2, the two
// lanes of a pair in the halves of one dword) against a lane-private dword column and against plain stores.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void rate(long long* cyc, unsigned* sink, int iters)
{
    __shared__ unsigned s_h[64 * 264];                     // (modes 5-7 use 64 x 256)
    for (int i = threadIdx.x; i < 64 * 264; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned x = (threadIdx.x + 1) * 2654435761u;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const unsigned c = (x >> (3 * n)) & 63u;
            if (MODE == 0) atomicAdd(&s_h[c * 136 + 32 * (wave & 3) + (lane >> 1)], 1u << (16 * (lane & 1)));      // the walk's pattern
            if (MODE == 1) atomicAdd(&s_h[c * 264 + 64 * (wave & 3) + lane], 1u);                                  // a dword per lane
            if (MODE == 2) s_h[c * 264 + 64 * (wave & 3) + lane] = x;                                              // plain store
            if (MODE == 3) atomicAdd(&s_h[c * 137 + 32 * (wave & 3) + (lane >> 1)], 1u << (16 * (lane & 1)));      // odd row stride
            if (MODE == 4) atomicAdd(&s_h[(c * 136 + 32 * (wave & 3) + (lane >> 1)) ^ (c >> 3)], 1u << (16 * (lane & 1)));
            if (MODE == 5) atomicAdd(&s_h[c * 256 + 64 * (wave & 3) + lane], 1u);                                  // a dword per lane, rows 256 dwords apart: bank = lane
            if (MODE == 6) atomicAdd(&s_h[c * 128 + 32 * (wave & 3) + (lane >> 1)], 1u << (16 * (lane & 1)));      // pairs, rows 128 dwords apart
            if (MODE == 7) atomicAdd(&s_h[c * 128 + 32 * (wave & 3) + (lane & 31)], 1u << (16 * (lane >> 5)));     // lanes l and l + 32 share a dword
        }
        x = x * 1664525u + 1013904223u;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    if (threadIdx.x == 0) sink[blockIdx.x] = s_h[x & 1023];
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    long long *d_cyc, *h = (long long*)malloc(cus * 16 * sizeof(long long));
    unsigned* sink;
    (void)hipMalloc(&d_cyc, cus * 16 * sizeof(long long));
    (void)hipMalloc(&sink, cus * sizeof(unsigned));
    const int iters = 400;
    const char* names[8] = {"walk pattern (pairs share a dword)", "a dword per lane", "plain stores", "pairs, row stride 137", "pairs, rows swizzled",
                            "a dword per lane, row stride 256", "pairs, row stride 128", "lanes l, l+32 share, stride 128"};
    for (int mode = 0; mode < 8; ++mode)
        for (int waves = 1; waves <= 16; waves *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 4) hipLaunchKernelGGL(rate<4>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 5) hipLaunchKernelGGL(rate<5>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 6) hipLaunchKernelGGL(rate<6>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                if (mode == 7) hipLaunchKernelGGL(rate<7>, dim3(cus), dim3(64 * waves), 0, 0, d_cyc, sink, iters);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, d_cyc, cus * 16 * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < cus; ++i)
                for (int w = 0; w < waves; ++w) s += (double)h[i * 16 + w];
            s /= (double)cus * waves;
            printf("%-36s waves/CU %2d : %6.1f clocks per LDS instruction of a wave, %6.1f per instruction of the CU\n", names[mode], waves,
                   s / (iters * 8.0), s / (iters * 8.0) / waves);
        }
    return 0;
}
