// dig_join.hip -- mutation x element-block interval join (the GPU side of the observed-count tabulation).
//
// Reference: tabulate_muts_per_sample_per_element (data_tools/mutation_tools.py:191-230) shells out to
// `bedtools intersect -wa -wb` on the mutation file and the bed6 blocks of the elements.  bedtools' test is the
// half-open overlap  m.start < b.end  and  b.start < m.end  on the same chromosome.
//
// Layout: blocks sorted by (chrom, start) and described by three int64 arrays of COMPOSITE keys
//     blk_start_key[i]  = chrom << 40 | start_i
//     blk_runmax_key[i] = chrom << 40 | max(end_0 .. end_i within the chromosome)   (non-decreasing overall)
//     blk_end[i]        = end_i
// so that for a mutation (chrom, s, e) the candidate blocks are the index range
//     lo = upper_bound(blk_runmax_key, chrom << 40 | s)        first block whose running max end exceeds s
//     hi = lower_bound(blk_start_key,  chrom << 40 | e)        first block starting at or after e
// and the overlapping ones are the candidates with blk_end > s (blocks may overlap or nest, so the candidates
// are filtered, not assumed).  One thread per mutation; two launches (count, then fill after an exclusive scan of
// the counts) keep the output order deterministic: mutation-major, blocks ascending.
#include "dig_common.hpp"

namespace dig {

constexpr int kJoinBlock = 256;

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int64_t upper_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool FILL>
__global__ __launch_bounds__(kJoinBlock) void overlap_kernel(const int64_t* __restrict__ blk_start_key,
                                                             const int64_t* __restrict__ blk_runmax_key,
                                                             const int64_t* __restrict__ blk_end, int64_t n_blk,
                                                             const int64_t* __restrict__ mut_chrom,
                                                             const int64_t* __restrict__ mut_start,
                                                             const int64_t* __restrict__ mut_end, int64_t n_mut,
                                                             int32_t* __restrict__ counts,
                                                             const int64_t* __restrict__ offsets,
                                                             int32_t* __restrict__ pair_mut,
                                                             int32_t* __restrict__ pair_blk)
{
    const int64_t stride = (int64_t)gridDim.x * kJoinBlock;
    for (int64_t m = (int64_t)blockIdx.x * kJoinBlock + threadIdx.x; m < n_mut; m += stride) {
        const int64_t s = mut_start[m];
        int64_t e = mut_end[m];
        if (e <= s) e = s + 1;                                  // zero-length feature: tested as [s, s+1)
        const int64_t base = mut_chrom[m] << 40;
        const int64_t lo = upper_bound_i64(blk_runmax_key, n_blk, base | s);
        const int64_t hi = lower_bound_i64(blk_start_key, n_blk, base | e);
        int32_t c = 0;
        int64_t o = FILL ? offsets[m] : 0;
        for (int64_t b = lo; b < hi; ++b) {
            if (blk_end[b] > s) {
                if (FILL) {
                    pair_mut[o] = (int32_t)m;
                    pair_blk[o] = (int32_t)b;
                    ++o;
                }
                ++c;
            }
        }
        if (!FILL) counts[m] = c;
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_overlap_join_count(const int64_t* blk_start_key, const int64_t* blk_runmax_key, const int64_t* blk_end,
                           int64_t n_blk, const int64_t* mut_chrom, const int64_t* mut_start, const int64_t* mut_end,
                           int64_t n_mut, int32_t* counts, void* stream)
{
    DIG_REQUIRE(n_blk >= 0 && n_mut >= 0, "sizes >= 0");
    if (n_mut == 0) return DIG_OK;
    DIG_REQUIRE(mut_chrom && mut_start && mut_end && counts, "non-null mutation arrays");
    DIG_REQUIRE(n_blk == 0 || (blk_start_key && blk_runmax_key && blk_end), "non-null block arrays");
    hipLaunchKernelGGL(overlap_kernel<false>, dim3(grid_for(n_mut, kJoinBlock)), dim3(kJoinBlock), 0, (hipStream_t)stream,
                       blk_start_key, blk_runmax_key, blk_end, n_blk, mut_chrom, mut_start, mut_end, n_mut, counts,
                       (const int64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_overlap_join_fill(const int64_t* blk_start_key, const int64_t* blk_runmax_key, const int64_t* blk_end,
                          int64_t n_blk, const int64_t* mut_chrom, const int64_t* mut_start, const int64_t* mut_end,
                          int64_t n_mut, const int64_t* offsets, int32_t* pair_mut, int32_t* pair_blk, void* stream)
{
    DIG_REQUIRE(n_blk >= 0 && n_mut >= 0, "sizes >= 0");
    if (n_mut == 0 || n_blk == 0) return DIG_OK;
    DIG_REQUIRE(mut_chrom && mut_start && mut_end && offsets && pair_mut && pair_blk, "non-null arrays");
    DIG_REQUIRE(blk_start_key && blk_runmax_key && blk_end, "non-null block arrays");
    hipLaunchKernelGGL(overlap_kernel<true>, dim3(grid_for(n_mut, kJoinBlock)), dim3(kJoinBlock), 0, (hipStream_t)stream,
                       blk_start_key, blk_runmax_key, blk_end, n_blk, mut_chrom, mut_start, mut_end, n_mut,
                       (int32_t*)nullptr, offsets, pair_mut, pair_blk);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
