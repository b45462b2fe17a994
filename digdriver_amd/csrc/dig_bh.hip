// dig_bh.hip -- Benjamini-Hochberg q-values of an ascending p-value list (the per-base route's last step).
//
// nb_model.get_q_vals (nb_model.py:340-342) = statsmodels.stats.multitest.fdrcorrection(pvals)[1], method 'indep':
//     q_(i) = min(1, min_{j >= i} p_(j) / (j / n))        over the p-values in ascending order, j = 1 .. n
// The sort stays torch's (rocPRIM radix sort: 0.6 ms per 7.2 M doubles); everything behind it -- the division by the empirical
// CDF, the REVERSE RUNNING MINIMUM and the cap -- is this one pass.  (torch.cummin, which the device form of get_q_vals used
// until round 5, is a generic scan that takes 21 ms per 7.2 M values on gfx950: 97 % of a cohort's q-values and 99 % of the whole
// per-base route of BASELINE configs[4], bench.py's aux_rooflines.)  Same IEEE operations in the same order as the host
// form -- p / (rank / n), np.minimum.accumulate from the end (a NaN makes everything in front of it NaN, as there), np.minimum(q, 1)
// -- so the same bits.  HBM-bound: 8 B read twice + 8 B written per value.
#include "dig_common.hpp"

namespace dig {

constexpr int kBhBlock = 256;
constexpr int kBhItems = 16;
constexpr int kBhChunk = kBhBlock * kBhItems;          // values per workgroup

__device__ __forceinline__ double nan_min(double a, double b)      // np.minimum: NaN if either is
{
    return (a != a) ? a : ((b != b) ? b : (b < a ? b : a));
}

__device__ __forceinline__ double bh_value(const double* __restrict__ ps, int64_t j, double n_f)
{
#pragma clang fp contract(off)
    return ps[j] / ((double)(j + 1) / n_f);
}

// reverse inclusive running minimum over the workgroup's chunk: thread t holds items [t * kBhItems, (t + 1) * kBhItems) of the chunk
// (blockIdx.y: the row of a batch -- one cohort's sorted p-values -- rows n apart)
template <bool WRITE>
__global__ __launch_bounds__(kBhBlock) void bh_chunk_kernel(const double* __restrict__ ps, int64_t n, double* __restrict__ chunk_min,
                                                            const double* __restrict__ suffix, double* __restrict__ q)
{
    const int64_t n_chunks_ = (n + kBhChunk - 1) / kBhChunk;
    ps += (int64_t)blockIdx.y * n;
    if (WRITE) {
        q += (int64_t)blockIdx.y * n;
        suffix += (int64_t)blockIdx.y * n_chunks_;
    } else
        chunk_min += (int64_t)blockIdx.y * n_chunks_;
    __shared__ double s_tot[kBhBlock];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const double n_f = (double)n;
    const int64_t base = (int64_t)blockIdx.x * kBhChunk + (int64_t)threadIdx.x * kBhItems;
    double v[kBhItems];
#pragma unroll
    for (int k = 0; k < kBhItems; ++k) v[k] = base + k < n ? bh_value(ps, base + k, n_f) : inf;
#pragma unroll
    for (int k = kBhItems - 2; k >= 0; --k) v[k] = nan_min(v[k], v[k + 1]);      // thread-local, from the thread's last item down
    s_tot[threadIdx.x] = v[0];
    __syncthreads();
    // reverse inclusive scan of the threads' totals (Hillis-Steele over 256 values)
    for (int d = 1; d < kBhBlock; d <<= 1) {
        const double mine = s_tot[threadIdx.x];
        const double other = (int)threadIdx.x + d < kBhBlock ? s_tot[threadIdx.x + d] : inf;
        __syncthreads();
        s_tot[threadIdx.x] = nan_min(mine, other);
        __syncthreads();
    }
    if (!WRITE) {
        if (threadIdx.x == 0) chunk_min[blockIdx.x] = s_tot[0];
        return;
    }
    double behind = (int)threadIdx.x + 1 < kBhBlock ? s_tot[threadIdx.x + 1] : inf;      // the threads after this one ...
    behind = nan_min(behind, suffix[blockIdx.x]);                                         // ... and the chunks after this one
#pragma unroll
    for (int k = 0; k < kBhItems; ++k)
        if (base + k < n) {
            const double m = nan_min(v[k], behind);
            q[base + k] = (m != m) ? m : (m < 1.0 ? m : 1.0);                            // np.minimum(q, 1.0)
        }
}

// suffix[b] = min of chunk_min[b + 1 ..] (inf for the last chunk), one workgroup, chunks walked from the end
__global__ __launch_bounds__(kBhBlock) void bh_suffix_kernel(const double* __restrict__ chunk_min, int64_t n_chunks, double* __restrict__ suffix)
{
    chunk_min += (int64_t)blockIdx.x * n_chunks;      // (one workgroup per row of the batch)
    suffix += (int64_t)blockIdx.x * n_chunks;
    __shared__ double s_tot[kBhBlock];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double carry = inf;                                   // minimum of everything behind the current stretch of kBhBlock chunks
    for (int64_t hi = n_chunks; hi > 0; hi -= kBhBlock) {
        const int64_t lo = hi - kBhBlock;                 // this stretch: chunks [lo, hi), thread t <-> chunk lo + t (may be < 0)
        const int64_t c = lo + threadIdx.x;
        s_tot[threadIdx.x] = c >= 0 ? chunk_min[c] : inf;
        __syncthreads();
        for (int d = 1; d < kBhBlock; d <<= 1) {
            const double mine = s_tot[threadIdx.x];
            const double other = (int)threadIdx.x + d < kBhBlock ? s_tot[threadIdx.x + d] : inf;
            __syncthreads();
            s_tot[threadIdx.x] = nan_min(mine, other);
            __syncthreads();
        }
        const double after = (int)threadIdx.x + 1 < kBhBlock ? s_tot[threadIdx.x + 1] : inf;
        if (c >= 0) suffix[c] = nan_min(after, carry);
        const double all = s_tot[0];
        __syncthreads();
        carry = nan_min(carry, all);
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int64_t dig_bh_workspace(int64_t n, int64_t rows)
{
    if (n <= 0 || rows <= 0) return 0;
    return 2 * rows * ((n + kBhChunk - 1) / kBhChunk) * (int64_t)sizeof(double);
}

int dig_bh_qvalues_sorted(const double* p_sorted, int64_t n, int64_t rows, double* q_sorted, void* workspace, int64_t workspace_bytes,
                          void* stream)
{
    DIG_REQUIRE(n >= 0 && rows >= 0, "n, rows >= 0");
    if (n == 0 || rows == 0) return DIG_OK;
    DIG_REQUIRE(p_sorted && q_sorted && workspace && workspace_bytes >= dig_bh_workspace(n, rows),
                "non-null pointers, workspace of dig_bh_workspace(n, rows) bytes");
    const int64_t n_chunks = (n + kBhChunk - 1) / kBhChunk;
    DIG_REQUIRE(n_chunks <= 0x7fffffff && rows <= 65535, "n or rows too large for one launch");
    double* chunk_min = (double*)workspace;
    double* suffix = chunk_min + rows * n_chunks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((bh_chunk_kernel<false>), dim3((unsigned)n_chunks, (unsigned)rows), dim3(kBhBlock), 0, s, p_sorted, n, chunk_min, nullptr, nullptr);
    hipLaunchKernelGGL(bh_suffix_kernel, dim3((unsigned)rows), dim3(kBhBlock), 0, s, chunk_min, n_chunks, suffix);
    hipLaunchKernelGGL((bh_chunk_kernel<true>), dim3((unsigned)n_chunks, (unsigned)rows), dim3(kBhBlock), 0, s, p_sorted, n, nullptr, suffix, q_sorted);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
