// Probe (developer tool): what a second wave of the same SIMD costs a chain of v_mfma_f64_16x16x4_f64, and the other way round.
// One workgroup of eight waves per CU: waves 0-3 issue NACC dependent chains of matrix instructions, waves 4-7 run (mode)
//   0 nothing, 1 integer vector work (bit-field extracts and adds), 2 the same + one LDS atomic per eight instructions,
//   3 FP64 vector FMAs.
// Printed: cycles per matrix instruction of a wave, cycles per vector instruction of a wave, each alone and side by side.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int mode>
__global__ __launch_bounds__(512) void mix(long long* cyc, double* sink, int it_m, int it_v, double a, double b)
{
    __shared__ unsigned s_h[64 * 136];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 136; i += 512) s_h[i] = 0;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (wave < 4) {
        double4_t acc[NACC];
        for (int n = 0; n < NACC; ++n) acc[n] = double4_t{0, 0, 0, 0};
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
        }
        double s = 0;
        for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (mode == 1 || mode == 2) {
        unsigned x = threadIdx.x * 2654435761u, sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {                      // two vector instructions per n (bit-field extract, multiply-add), eight independent chains
                const unsigned c = (x >> (2 * n)) & 63u;
                if (mode == 2) atomicAdd(&s_h[c * 136 + ((threadIdx.x & 255) >> 1)], 1u << (16 * (threadIdx.x & 1)));
                else sum[n] += c * 136u;
            }
            x = (x << 1) ^ (x >> 3);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7];
    } else if (mode == 3) {
        double v[8];
        for (int n = 0; n < 8; ++n) v[n] = threadIdx.x + n;
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int n = 0; n < 8; ++n) v[n] = fma(v[n], a, b);
        }
        double s = 0;
        for (int n = 0; n < 8; ++n) s += v[n];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int mode>
static void run_mode(int cus, long long* d_cyc, double* sink, long long* h)
{
        for (int with_m = (mode == 0 ? 1 : 0); with_m <= 1; ++with_m) {
            const int it_m = with_m ? 8000 / NACC : 0;
            const int it_v = mode == 0 ? 0 : (mode == 2 ? 500 : 2000);     // (the vector role ends well inside the matrix role)
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL((mix<NACC, mode>), dim3(cus), dim3(512), 0, 0, d_cyc, sink, it_m, it_v, 1.0, 1e-9);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, d_cyc, cus * 8 * sizeof(long long), hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            for (int i = 0; i < cus; ++i)
                for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[i * 8 + w];
            m /= cus * 4.0;
            v /= cus * 4.0;
            printf("chains %d  vector mode %d  matrix %s : %8.1f clocks per matrix instr, %7.2f clocks per vector instr (mode 2: per atomic + 2 vector instr)\n", NACC,
                   mode, with_m ? "on " : "off", with_m ? m / (it_m * NACC) : 0.0, mode ? v / (it_v * (mode == 3 ? 8.0 : mode == 2 ? 8.0 : 19.0)) : 0.0);
        }
}

template <int NACC>
static void run(int cus, long long* d_cyc, double* sink)
{
    long long* h = (long long*)malloc(cus * 8 * sizeof(long long));
    run_mode<NACC, 0>(cus, d_cyc, sink, h);
    run_mode<NACC, 1>(cus, d_cyc, sink, h);
    run_mode<NACC, 2>(cus, d_cyc, sink, h);
    run_mode<NACC, 3>(cus, d_cyc, sink, h);
    free(h);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    long long* d_cyc;
    double* sink;
    (void)hipMalloc(&d_cyc, cus * 8 * sizeof(long long));
    (void)hipMalloc(&sink, cus * 512 * sizeof(double));
    printf("CUs %d\n", cus);
    run<1>(cus, d_cyc, sink);
    run<2>(cus, d_cyc, sink);
    run<4>(cus, d_cyc, sink);
    return 0;
}
