// dig_suffstats.hip -- per-cohort sufficient statistics for the cohort scale factors.
//
// Reference: calc_scale_factor_efficient, genome mode (driver_model/transfer_tools.py:148-156):
//     regions_pass = regions[~regions.FLAG];  N_SNV_EXP = regions_pass.Y_PRED.sum()
//     cj_snv = N_SNV_OBS / N_SNV_EXP;  cj_ind = N_IND_OBS / N_SNV_EXP
// For C cohorts at once this is a column reduction of the [N, C] rate table with the FLAG mask.
//
// Two deterministic stages (fixed summation order -> bit-reproducible run to run and, with the
// rank-ordered sum of the all-gathered partials, across GPU counts):
//   stage 1: every workgroup owns a contiguous block of rows; thread t owns column t % C and
//            rows r0 + t / C, r0 + t / C + rpp, ... so a pass over rpp rows is one contiguous,
//            fully coalesced run of rpp * C doubles; per-thread sums are combined through LDS in
//            row-group order and written to partial[workgroup][C];
//   stage 2: one workgroup per column adds the partials with a fixed-shape tree.
#include "dig_common.hpp"

namespace dig {

constexpr int kSsBlock = 256;

__global__ __launch_bounds__(kSsBlock) void suffstats_stage1(const double* __restrict__ bin_mu,
                                                             const uint8_t* __restrict__ bin_flag, int64_t N,
                                                             int64_t C, int64_t rows_per_block,
                                                             double* __restrict__ partial)
{
    __shared__ double part[kSsBlock];
    const int tid = threadIdx.x;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < N) ? r_begin + rows_per_block : N;
    if (C <= kSsBlock) {
        const int rpp = kSsBlock / (int)C;              // rows per pass
        const int col = tid % (int)C, rg = tid / (int)C;
        double acc = 0.0;
        if (rg < rpp) {
            int64_t r = r_begin + rg;
            // 4 independent loads in flight per thread
            for (; r + 3 * rpp < r_end; r += 4 * rpp) {
                const int64_t o0 = r * C + col, o1 = o0 + rpp * C, o2 = o1 + rpp * C, o3 = o2 + rpp * C;
                const double v0 = bin_mu[o0], v1 = bin_mu[o1], v2 = bin_mu[o2], v3 = bin_mu[o3];
                const uint8_t f0 = bin_flag[o0], f1 = bin_flag[o1], f2 = bin_flag[o2], f3 = bin_flag[o3];
                acc += f0 ? 0.0 : v0;
                acc += f1 ? 0.0 : v1;
                acc += f2 ? 0.0 : v2;
                acc += f3 ? 0.0 : v3;
            }
            for (; r < r_end; r += rpp) {
                const int64_t o = r * C + col;
                acc += bin_flag[o] ? 0.0 : bin_mu[o];
            }
        }
        part[tid] = acc;
        __syncthreads();
        if (tid < C) {
            double s = 0.0;
            for (int g = 0; g < rpp; ++g) s += part[g * (int)C + tid];
            partial[(int64_t)blockIdx.x * C + tid] = s;
        }
    } else {
        for (int64_t c = tid; c < C; c += kSsBlock) {
            double acc = 0.0;
            for (int64_t r = r_begin; r < r_end; ++r) {
                const int64_t o = r * C + c;
                acc += bin_flag[o] ? 0.0 : bin_mu[o];
            }
            partial[(int64_t)blockIdx.x * C + c] = acc;
        }
    }
}

// stage 2: one workgroup per cohort column; thread t adds partial[t], partial[t + 256], ... (independent
// loads), then a fixed-shape LDS tree combines the 256 thread sums -> deterministic.
__global__ __launch_bounds__(kSsBlock) void suffstats_stage2(const double* __restrict__ partial, int nblocks, int64_t C,
                                                             double* __restrict__ out)
{
    __shared__ double red[kSsBlock];
    const int64_t c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kSsBlock) s += partial[(int64_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kSsBlock / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

static int ss_blocks(int64_t N)
{
    int64_t g = (int64_t)cu_count() * 4;
    if (g > (N + 63) / 64) g = (N + 63) / 64;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace dig

using namespace dig;

extern "C" {

int64_t dig_scale_suffstats_workspace(int64_t N, int64_t C)
{
    if (N <= 0 || C <= 0) return 0;
    return (int64_t)ss_blocks(N) * C * (int64_t)sizeof(double);
}

int dig_scale_suffstats(const double* bin_mu, const uint8_t* bin_flag, int64_t N, int64_t C, double* out_sum,
                        void* workspace, int64_t workspace_bytes, void* stream)
{
    DIG_REQUIRE(N >= 0 && C >= 0, "N, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(out_sum, "non-null output");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        DIG_HIP_TRY(hipMemsetAsync(out_sum, 0, (size_t)C * sizeof(double), s));
        return DIG_OK;
    }
    DIG_REQUIRE(bin_mu && bin_flag && workspace, "non-null inputs and workspace");
    const int g = ss_blocks(N);
    DIG_REQUIRE(workspace_bytes >= (int64_t)g * C * (int64_t)sizeof(double), "workspace smaller than dig_scale_suffstats_workspace(N, C)");
    DIG_REQUIRE(((uintptr_t)workspace & 7u) == 0, "workspace 8-byte aligned");
    const int64_t rpb = (N + g - 1) / g;
    hipLaunchKernelGGL(suffstats_stage1, dim3(g), dim3(kSsBlock), 0, s, bin_mu, bin_flag, N, C, rpb, (double*)workspace);
    DIG_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(suffstats_stage2, dim3((unsigned)C), dim3(kSsBlock), 0, s, (const double*)workspace, g, C, out_sum);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_scale_suffstats_host(const double* bin_mu, const uint8_t* bin_flag, int64_t N, int64_t C, double* out_sum,
                             int device)
{
    DIG_REQUIRE(N >= 0 && C >= 0, "N, C >= 0");
    if (C == 0) return DIG_OK;
    DIG_REQUIRE(out_sum, "non-null output");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dmu, dfl, dws, dout;
    const size_t n = (size_t)N * C;
    DIG_HIP_TRY(dmu.alloc(n * 8));
    DIG_HIP_TRY(dfl.alloc(n));
    DIG_HIP_TRY(dws.alloc((size_t)dig_scale_suffstats_workspace(N, C)));
    DIG_HIP_TRY(dout.alloc((size_t)C * 8));
    if (n) {
        DIG_REQUIRE(bin_mu && bin_flag, "non-null inputs");
        DIG_HIP_TRY(hipMemcpy(dmu.p, bin_mu, n * 8, hipMemcpyHostToDevice));
        DIG_HIP_TRY(hipMemcpy(dfl.p, bin_flag, n, hipMemcpyHostToDevice));
    }
    int rc = dig_scale_suffstats(dmu.as<double>(), dfl.as<uint8_t>(), N, C, dout.as<double>(), dws.p,
                                 dig_scale_suffstats_workspace(N, C), nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out_sum, dout.p, (size_t)C * 8, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
