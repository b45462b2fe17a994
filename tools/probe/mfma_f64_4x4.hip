// Probe: lane layout and issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (developer tool).
// One-hot experiment: a = 1 in lane p only, b = 1 in lane q only; the lane(s) of D that read 1 tell which (block, i, k)
// lane p feeds and which (block, k, j) lane q feeds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void onehot(int* out)           // grid 64 x 64: out[p][q] = bit mask of lanes with D != 0 (as two ints) + value
{
    const int p = blockIdx.x, q = blockIdx.y, l = threadIdx.x;
    const double a = l == p ? 1.0 : 0.0, b = l == q ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    const unsigned long long m = __ballot(d != 0.0);
    if (l == 0) { out[(p * 64 + q) * 2] = (int)(m & 0xffffffffu); out[(p * 64 + q) * 2 + 1] = (int)(m >> 32); }
}
template <int CHAINS>
__global__ void rate(double* out, int n)
{
    double acc[CHAINS];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)n * CHAINS);
}
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void rate16(double* out, int n)
{
    double4_t acc[CHAINS];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int c = 0; c < CHAINS; ++c) acc[c] = double4_t{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)n * CHAINS);
}
int main()
{
    int* dout; hipMalloc(&dout, 64 * 64 * 2 * sizeof(int));
    hipLaunchKernelGGL(onehot, dim3(64, 64), dim3(64), 0, 0, dout);
    std::vector<int> h(64 * 64 * 2);
    hipMemcpy(h.data(), dout, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    // for every a-lane p: which b-lanes q give a product, and where it lands
    for (int p = 0; p < 64; ++p) {
        printf("a lane %2d:", p);
        for (int q = 0; q < 64; ++q) {
            unsigned long long m = ((unsigned long long)(unsigned)h[(p * 64 + q) * 2 + 1] << 32) | (unsigned)h[(p * 64 + q) * 2];
            if (m) { printf("  b%2d->d", q); for (int l = 0; l < 64; ++l) if ((m >> l) & 1) printf("%d,", l); }
        }
        printf("\n");
    }
    double* dd; hipMalloc(&dd, 1 << 20);
    double v;
    hipLaunchKernelGGL((rate<1>), dim3(1), dim3(64), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("4x4x4: 1 wave, 1 dependent chain: %.1f cycles per MFMA\n", v);
    hipLaunchKernelGGL((rate<4>), dim3(1), dim3(64), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("4x4x4: 1 wave, 4 chains: %.1f cycles per MFMA\n", v);
    hipLaunchKernelGGL((rate<4>), dim3(1), dim3(256), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("4x4x4: 4 waves (1 per SIMD), 4 chains: %.1f cycles per MFMA\n", v);
    hipLaunchKernelGGL((rate<4>), dim3(1), dim3(512), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("4x4x4: 8 waves (2 per SIMD), 4 chains: %.1f cycles per MFMA per wave\n", v);
    hipLaunchKernelGGL((rate16<1>), dim3(1), dim3(64), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("16x16x4: 1 wave, 1 chain: %.1f cycles per MFMA\n", v);
    hipLaunchKernelGGL((rate16<3>), dim3(1), dim3(64), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("16x16x4: 1 wave, 3 chains: %.1f cycles per MFMA\n", v);
    hipLaunchKernelGGL((rate16<3>), dim3(1), dim3(512), 0, 0, dd, 10000); hipMemcpy(&v, dd, 8, hipMemcpyDeviceToHost); printf("16x16x4: 8 waves, 3 chains: %.1f cycles per MFMA per wave\n", v);
    return 0;
}
