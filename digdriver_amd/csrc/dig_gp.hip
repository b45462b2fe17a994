// dig_gp.hip -- the two elementwise passes of the sparse-GP calibration's cross-covariance on gfx950.
//
// Reference: the RBF kernel of the SGPR (gp_trainer.py:28-45: ScaleKernel(RBFKernel) between the m inducing points and
// the n <= 150 000 training rows) and its gradient.  With the squared distances taken from one GEMM (G = Z X^T),
// K[i][j] = os exp(c max(|z_i|^2 + |x_j|^2 - 2 G[i][j], 0)),  c = -1 / (2 l^2),  is ONE pass over the m x n matrix, and
// so is everything its backward needs:  W = g o K,  the row sums of W (for dZ = (W X - rowsum(W) o Z) / l^2), sum(W)
// (d os = sum(W) / os) and sum(W d2) (d l = sum(W d2) / l^3).  Built from torch's elementwise operators the same work
// took about ten passes of 0.96 GB each per direction: half of the fit's time (rocprofv3: 474 launches per Adam step).
// HBM-bound: 16 B per element forward (read G, write K in place), 24 B backward (read g and K, write W).
#include "dig_common.hpp"

namespace dig {

constexpr int kGpBlock = 256;
constexpr int kGpPerThread = 4;
constexpr int kGpChunk = kGpBlock * kGpPerThread;          // columns per workgroup

__global__ __launch_bounds__(kGpBlock) void rbf_from_gram_kernel(double* __restrict__ G, const double* __restrict__ a2,
                                                                 const double* __restrict__ b2, int64_t n, double c, double os)
{
    const int64_t row = blockIdx.y;
    const double ai = a2[row];
    double* g = G + row * n;
    const int64_t j0 = (int64_t)blockIdx.x * kGpChunk + threadIdx.x;
#pragma unroll
    for (int u = 0; u < kGpPerThread; ++u) {
        const int64_t j = j0 + (int64_t)u * kGpBlock;
        if (j < n) {
            const double d2 = fmax(ai + b2[j] - 2.0 * g[j], 0.0);
            g[j] = os * exp(c * d2);
        }
    }
}

// partial[row][chunk][0..2] = sum W, sum W d2, (unused); d2 is recovered from K: d2 = log(K / os) / c (K == 0: W == 0)
__global__ __launch_bounds__(kGpBlock) void rbf_backward_kernel(const double* __restrict__ g, const double* __restrict__ K,
                                                                double* __restrict__ W, double* __restrict__ partial,
                                                                int64_t n, double inv_c, double inv_os)
{
    __shared__ double s_w[kGpBlock / 64], s_wd[kGpBlock / 64];
    const int64_t row = blockIdx.y;
    const int64_t j0 = (int64_t)blockIdx.x * kGpChunk + threadIdx.x;
    double sw = 0.0, swd = 0.0;
#pragma unroll
    for (int u = 0; u < kGpPerThread; ++u) {
        const int64_t j = j0 + (int64_t)u * kGpBlock;
        if (j < n) {
            const double k = K[row * n + j];
            const double w = g[row * n + j] * k;
            W[row * n + j] = w;
            sw += w;
            if (k > 0.0) swd += w * (log(k * inv_os) * inv_c);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sw += __shfl_xor(sw, o, 64);
        swd += __shfl_xor(swd, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = sw;
        s_wd[threadIdx.x >> 6] = swd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kGpBlock / 64; ++w) {                  // fixed order: deterministic
            a += s_w[w];
            b += s_wd[w];
        }
        double* p = partial + (row * gridDim.x + blockIdx.x) * 2;
        p[0] = a;
        p[1] = b;
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_rbf_from_gram(double* G, const double* a2, const double* b2, int64_t m, int64_t n, double lengthscale,
                      double outputscale, void* stream)
{
    DIG_REQUIRE(m >= 0 && n >= 0 && m < 65536, "0 <= m < 65536, n >= 0");
    DIG_REQUIRE(lengthscale > 0.0 && outputscale > 0.0, "positive lengthscale and outputscale");
    if (m == 0 || n == 0) return DIG_OK;
    DIG_REQUIRE(G && a2 && b2, "non-null pointers");
    const dim3 grid((unsigned)((n + kGpChunk - 1) / kGpChunk), (unsigned)m);
    hipLaunchKernelGGL(rbf_from_gram_kernel, grid, dim3(kGpBlock), 0, (hipStream_t)stream, G, a2, b2, n,
                       -0.5 / (lengthscale * lengthscale), outputscale);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int64_t dig_rbf_backward_partials(int64_t m, int64_t n)
{
    if (m < 0 || n < 0) return -1;
    return m * ((n + kGpChunk - 1) / kGpChunk) * 2;
}

int dig_rbf_backward(const double* g, const double* K, int64_t m, int64_t n, double lengthscale, double outputscale, double* W,
                     double* partial, void* stream)
{
    DIG_REQUIRE(m >= 0 && n >= 0 && m < 65536, "0 <= m < 65536, n >= 0");
    DIG_REQUIRE(lengthscale > 0.0 && outputscale > 0.0, "positive lengthscale and outputscale");
    if (m == 0 || n == 0) return DIG_OK;
    DIG_REQUIRE(g && K && W && partial, "non-null pointers");
    const dim3 grid((unsigned)((n + kGpChunk - 1) / kGpChunk), (unsigned)m);
    hipLaunchKernelGGL(rbf_backward_kernel, grid, dim3(kGpBlock), 0, (hipStream_t)stream, g, K, W, partial, n,
                       -2.0 * lengthscale * lengthscale, 1.0 / outputscale);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
