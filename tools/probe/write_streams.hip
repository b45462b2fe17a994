// Probe: does the number of concurrently written arrays matter for the write rate of a streaming kernel on MI355X?
//   hipcc -O3 --offload-arch=gfx950 write_streams.hip -o write_streams && ./write_streams
// A persistent grid (one 1024-thread workgroup per CU, tiles of 64 items in XCD-contiguous runs like the statistics
// kernel) writes 13 doubles per item (a) as 13 separate arrays, 512 contiguous bytes per wave and array, (b) as one
// array of 13-double records through an LDS transpose (every store instruction still writes 512 contiguous bytes),
// and reads 5 doubles per item from 5 arrays in both cases.  Prints the achieved GB/s (read + written).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kOut = 13, kIn = 5;

template <bool RECORDS, bool NT = true>
__global__ __launch_bounds__(1024) void writer(const double* __restrict__ in, double* __restrict__ out, long n)
{
    __shared__ unsigned s_ticket;
    __shared__ double s_tr[16][64 * kOut];
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n_tiles = n / 64;
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        const long tile = ((long)t * 8 + (blockIdx.x & 7)) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (tile >= n_tiles) break;
        const long i = tile * 64 + lane;
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < kIn; ++j) v += in[(long)j * n + i];
        if (!RECORDS) {
#pragma unroll
            for (int j = 0; j < kOut; ++j) { if (NT) __builtin_nontemporal_store(v + j, &out[(long)j * n + i]); else out[(long)j * n + i] = v + j; }
        } else {
            double* tr = s_tr[wave];
#pragma unroll
            for (int j = 0; j < kOut; ++j) tr[lane * kOut + j] = v + j;       // record-major in LDS
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double* dst = out + tile * 64 * kOut;                              // 64 records = 6 656 contiguous bytes
#pragma unroll
            for (int j = 0; j < kOut; ++j) __builtin_nontemporal_store(tr[j * 64 + lane], &dst[j * 64 + lane]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 120091L * 37 / 64 * 64;      // (a second run with 4443367 = E*C itself: planes not 64-byte aligned)
    double *in, *out;
    hipMalloc(&in, sizeof(double) * kIn * n);
    hipMalloc(&out, sizeof(double) * kOut * n);
    hipMemset(in, 0, sizeof(double) * kIn * n);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const double bytes = 8.0 * (kIn + kOut) * n;
    for (int rep = 0; rep < 3; ++rep)
        for (int form = 0; form < 3; ++form) {
            for (int w = 0; w < 20; ++w)
                if (form == 2) hipLaunchKernelGGL((writer<false, false>), dim3(grid), dim3(1024), 0, 0, in, out, n);
                else if (form) hipLaunchKernelGGL(writer<true>, dim3(grid), dim3(1024), 0, 0, in, out, n);
                else hipLaunchKernelGGL(writer<false>, dim3(grid), dim3(1024), 0, 0, in, out, n);
            hipEventRecord(a, 0);
            const int K = 100;
            for (int w = 0; w < K; ++w)
                if (form == 2) hipLaunchKernelGGL((writer<false, false>), dim3(grid), dim3(1024), 0, 0, in, out, n);
                else if (form) hipLaunchKernelGGL(writer<true>, dim3(grid), dim3(1024), 0, 0, in, out, n);
                else hipLaunchKernelGGL(writer<false>, dim3(grid), dim3(1024), 0, 0, in, out, n);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("%s: %.1f us per launch, %.0f GB/s (%.0f MB moved)\n", form == 2 ? "13 separate arrays, plain stores" : form ? "one record array (LDS transpose)" : "13 separate arrays            ",
                   ms / K * 1e3, bytes / (ms / K * 1e-3) / 1e9, bytes / 1e6);
        }
    return 0;
}
