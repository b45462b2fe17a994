// dig_gp.hip -- the two elementwise passes of the sparse-GP calibration's cross-covariance on gfx950.
//
// Reference: the RBF kernel of the SGPR (gp_trainer.py:28-45: ScaleKernel(RBFKernel) between the m inducing points and
// the n <= 150 000 training rows) and its gradient.  K[i][j] = os exp(c |z_i - x_j|^2),  c = -1 / (2 l^2),  is written in
// ONE pass over the m x n matrix straight from the points (a K = 16 GEMM for the Gram matrix alone took 1.2 ms: it is
// bound by its 0.5 GB of output), and one pass is everything its backward needs:  W = g o K,  the row sums of W (for dZ = (W X - rowsum(W) o Z) / l^2), sum(W)
// (d os = sum(W) / os) and sum(W d2) (d l = sum(W d2) / l^3).  Built from torch's elementwise operators the same work
// took about ten passes of 0.96 GB each per direction: half of the fit's time (rocprofv3: 474 launches per Adam step).
// HBM-bound: 8 B per element forward (write K), 24 B backward (read g and K, write W).
#include "dig_common.hpp"

namespace dig {

constexpr int kGpBlock = 256;
constexpr int kGpPerThread = 4;
constexpr int kGpChunk = kGpBlock * kGpPerThread;          // columns per workgroup

// K[i][j] = os exp(c |z_i - x_j|^2) straight from the points (d <= 32 features; 16 in the reference): a thread keeps its
// column's x_j in registers and walks the rows of a 16-row group, whose z sit in LDS (broadcast reads).  The differences
// are formed directly -- no Gram matrix, no |z|^2 + |x|^2 - 2 z.x cancellation -- and the m x n matrix is written once.
constexpr int kGpMaxD = 32;
constexpr int kGpRows = 16;

template <int D>
__global__ __launch_bounds__(kGpBlock) void rbf_cross_kernel(const double* __restrict__ Z, const double* __restrict__ X,
                                                             double* __restrict__ K, int64_t m, int64_t n, double c, double os)
{
    __shared__ double s_z[kGpRows][D];
    const int64_t row0 = (int64_t)blockIdx.y * kGpRows;
    const int rows = (int)((m - row0) < kGpRows ? (m - row0) : kGpRows);
    for (int i = threadIdx.x; i < rows * D; i += kGpBlock) s_z[i / D][i % D] = Z[(row0 + i / D) * D + i % D];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * kGpBlock + threadIdx.x;
    if (j >= n) return;
    double x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = X[j * D + d];
    for (int r = 0; r < rows; ++r) {
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double t = s_z[r][d] - x[d];
            d2 = fma(t, t, d2);
        }
        __builtin_nontemporal_store(os * exp(c * d2), &K[(row0 + r) * n + j]);
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

int dig_rbf_cross(const double* Z, const double* X, int64_t m, int64_t n, int64_t d, double lengthscale, double outputscale,
                  double* K, void* stream)
{
    DIG_REQUIRE(m >= 0 && n >= 0 && m < 65536 * (int64_t)kGpRows, "0 <= m < 2^20, n >= 0");
    DIG_REQUIRE(d >= 1 && d <= kGpMaxD, "1 <= d <= 32 features");
    DIG_REQUIRE(lengthscale > 0.0 && outputscale > 0.0, "positive lengthscale and outputscale");
    if (m == 0 || n == 0) return DIG_OK;
    DIG_REQUIRE(Z && X && K, "non-null pointers");
    const dim3 grid((unsigned)((n + kGpBlock - 1) / kGpBlock), (unsigned)((m + kGpRows - 1) / kGpRows));
    const double c = -0.5 / (lengthscale * lengthscale);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(kGpBlock), 0, (hipStream_t)stream, Z, X, K, m, n, c, outputscale); };
    switch ((int)d) {                       // (the feature count is a compile-time constant of the inner loop)
#define DIG_GP_CASE(D) case D: go(rbf_cross_kernel<D>); break;
        DIG_GP_CASE(1) DIG_GP_CASE(2) DIG_GP_CASE(3) DIG_GP_CASE(4) DIG_GP_CASE(5) DIG_GP_CASE(6) DIG_GP_CASE(7) DIG_GP_CASE(8)
        DIG_GP_CASE(9) DIG_GP_CASE(10) DIG_GP_CASE(11) DIG_GP_CASE(12) DIG_GP_CASE(13) DIG_GP_CASE(14) DIG_GP_CASE(15) DIG_GP_CASE(16)
        DIG_GP_CASE(17) DIG_GP_CASE(18) DIG_GP_CASE(19) DIG_GP_CASE(20) DIG_GP_CASE(21) DIG_GP_CASE(22) DIG_GP_CASE(23) DIG_GP_CASE(24)
        DIG_GP_CASE(25) DIG_GP_CASE(26) DIG_GP_CASE(27) DIG_GP_CASE(28) DIG_GP_CASE(29) DIG_GP_CASE(30) DIG_GP_CASE(31) DIG_GP_CASE(32)
#undef DIG_GP_CASE
    }
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
