// dig_tiles.hip -- front half of the per-base / tiled negative-binomial route on gfx950.
//
// Reference (one pysam fetch, one tabix fetch and a Python loop over 10 000 positions per bin):
//   sequence_tools.py:292-317   base_probabilities_by_region: per position S_prob[context] (0 when the window holds a
//                               non-ACGT base), normalised over the region
//   nb_model.py:126-186         apply_nb_to_region: tiles of `binsize` positions, pt = sum of the tile's probabilities,
//                               k = mutation rows whose START is one of the tile's positions
//   nb_model.py:188-234         nb_model: all bins of a cohort
// for trinucleotide contexts (n_up = n_down = 1: the 64 / 192-type sequence model every live part of the pipeline uses).
// The back half -- p = 1 / (pt theta + 1), nb_pvalue_exact, exp = pt mu -- is dig_tiled_nb_test (dig_nb.hip).
//
// dig_base_tile_probs: one workgroup per region, all cohorts at once.  Per-position probabilities never exist:
//   1. the region's packed bases (0.5 B per base) go to LDS;
//   2. every lane owns one tile and counts its positions into a private 64-bin context histogram h[ctx][tile] (LDS,
//      16-bit counters, no atomics: the counter column belongs to the lane); the region histogram H = sum over tiles;
//   3. T[c] = sum_ctx H[ctx] S[c][ctx] (the reference's np.sum over positions, regrouped by context);
//   4. pt[c][tile] = (sum_ctx h[ctx][tile] S[c][ctx]) / T[c]: lane = tile, the 64 x C table is read as LDS broadcasts
//      (one address per wave), twenty cohort accumulators in registers per sweep of the histogram column, two cohorts per 16-byte LDS read.
// What bounds it: the FP64 multiply-adds of step 4 (2 x 64 x C flops per tile: 4.7 kflop at C = 37 against 25 bytes of
// genome) -- FP64 VALU, not HBM; the outputs are 8 C bytes per tile.  binsize == 1 (tile = position) skips the
// histograms: pt = S[c][ctx] / T[c].  Rounding differs from the reference's per-position normalise-then-sum by a few
// ulp (1e-15 relative, tests/test_gpu_tiles.py); counts are exact.
//
// dig_tile_mut_counts: k[c][region][tile] from the (mutation, region) pairs of dig_overlap_join_*: one atomic add per
// pair whose START lies inside the region's positions.
#include "dig_common.hpp"

namespace dig {

constexpr int kTileBlock = 256;
constexpr int kTileCohorts = 20;          // cohort accumulators per sweep of the histogram column (37 cohorts: 20 + 17); even
constexpr int kTileMaxWords = 1536;       // packed words staged per pass: 12 288 bases (a 10-kb bin and its neighbours)

struct TileRegion {
    int64_t first;      // first position (chromosome coordinates)
    int64_t n_pos;      // number of positions
    int64_t g0;         // global base index of position `first` (counted from word 1 of the genome array)
};

// fetch_sequence (sequence_tools.py:21-29) with n_up = n_down = 1: START == 0 becomes 1; the widened fetch is cut at
// the chromosome end, so the last position with a full window is chrom_len - 2.
__device__ __forceinline__ TileRegion tile_region(const int64_t* chrom_off, const int64_t* chrom_len, int chrom, int64_t start,
                                                  int64_t end)
{
    TileRegion t;
    const int64_t len = chrom_len[chrom];
    t.first = start == 0 ? 1 : start;
    const int64_t stop = end < len - 1 ? end : len - 1;       // one past the last position
    t.n_pos = stop > t.first ? stop - t.first : 0;
    t.g0 = chrom_off[chrom] + t.first;
    return t;
}

// 4-bit code of global base g (word 0 of the array is the leading pad word)
__device__ __forceinline__ unsigned tile_base(const uint32_t* s_words, int64_t g, int64_t g_lds0)
{
    const int64_t r = g - g_lds0;                               // base index inside the staged words
    return (s_words[r >> 3] >> (4 * (int)(r & 7))) & 15u;
}

template <bool SINGLE>
__global__ __launch_bounds__(kTileBlock) void base_tile_probs_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, int64_t R, const double* __restrict__ s_prob, int64_t C, int binsize, int64_t n_tiles,
    double* __restrict__ pt, int64_t* __restrict__ first_pos, int32_t* __restrict__ n_valid)
{
    __shared__ uint32_t s_words[kTileMaxWords + 2];
    __shared__ unsigned short s_hist[64][kTileBlock];          // h[ctx][tile of the chunk]
    __shared__ unsigned s_H[64];
    __shared__ __attribute__((aligned(16))) double s_S[64][kTileCohorts];      // [context][cohort of the group]: two cohorts per 16-byte LDS read
    __shared__ double s_T[kTileCohorts];
    const int tid = threadIdx.x;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const TileRegion reg = tile_region(chrom_off, chrom_len, reg_chrom[r], reg_start[r], reg_end[r]);
        const int64_t tiles_valid = (reg.n_pos + binsize - 1) / binsize;
        if (tid == 0) {
            first_pos[r] = reg.first;
            n_valid[r] = (int32_t)(tiles_valid < n_tiles ? tiles_valid : n_tiles);
        }
        // ---- region histogram H over ALL positions (passes of kTileMaxWords words when the region is longer) ----
        if (tid < 64) s_H[tid] = 0;
        __syncthreads();
        const int64_t pos_per_pass = (int64_t)(kTileMaxWords - 1) * 8;
        for (int64_t p0 = 0; p0 < reg.n_pos; p0 += pos_per_pass) {
            const int64_t np = reg.n_pos - p0 < pos_per_pass ? reg.n_pos - p0 : pos_per_pass;
            const int64_t ga = reg.g0 + p0 - 1;                 // leftmost base needed (left neighbour of the first position)
            const int64_t w0 = (ga >> 3) + 1;                   // array word holding it (array word = genome word + 1)
            const int64_t nw = ((ga + np + 1) >> 3) + 1 - w0 + 1;
            for (int64_t i = tid; i < nw; i += kTileBlock) s_words[i] = words[(w0 + i < n_words ? w0 + i : n_words - 1)];
            __syncthreads();
            const int64_t g_lds0 = (w0 - 1) << 3;
            for (int64_t j = tid; j < np; j += kTileBlock) {
                const int64_t g = reg.g0 + p0 + j;
                const unsigned a = tile_base(s_words, g - 1, g_lds0), b = tile_base(s_words, g, g_lds0), c = tile_base(s_words, g + 1, g_lds0);
                if (!((a | b | c) & 12u)) atomicAdd(&s_H[16 * a + 4 * b + c], 1u);
            }
            __syncthreads();
        }
        // ---- tiles in chunks of kTileBlock (lane = tile): histogram columns once, then the cohorts in groups ----
        for (int64_t t0 = 0; t0 < n_tiles; t0 += kTileBlock) {
            const int64_t t = t0 + tid;
            const bool live = t < tiles_valid && t < n_tiles;
            const int64_t pa = t0 * binsize;                    // first position of the chunk
            int64_t np = (int64_t)kTileBlock * binsize;
            if (np > reg.n_pos - pa) np = reg.n_pos - pa;
            if (np < 0) np = 0;
            unsigned ctx_single = 64;
            if (!SINGLE)
                for (int x = 0; x < 64; ++x) s_hist[x][tid] = 0;
            // 256 tiles span 256 * binsize positions: staged in passes of whole tiles when that exceeds the buffer
            const int64_t pos_pass = (int64_t)(kTileMaxWords - 1) * 8 / binsize * binsize;
            for (int64_t q0 = 0; q0 < np; q0 += pos_pass) {
                const int64_t nq = np - q0 < pos_pass ? np - q0 : pos_pass;
                const int64_t ga = reg.g0 + pa + q0 - 1;
                const int64_t w0 = (ga >> 3) + 1;
                const int64_t nw = ((ga + nq + 1) >> 3) + 1 - w0 + 1;
                __syncthreads();
                for (int64_t i = tid; i < nw; i += kTileBlock) s_words[i] = words[(w0 + i < n_words ? w0 + i : n_words - 1)];
                __syncthreads();
                const int64_t g_lds0 = (w0 - 1) << 3;
                const int64_t tp = (int64_t)tid * binsize - q0;          // first position of this lane's tile inside the pass
                if (live && tp >= 0 && tp < nq) {
                    int64_t cnt = binsize;
                    if (cnt > nq - tp) cnt = nq - tp;
                    const int64_t g = reg.g0 + pa + q0 + tp;
                    unsigned a = tile_base(s_words, g - 1, g_lds0), b = tile_base(s_words, g, g_lds0);
                    for (int64_t j = 0; j < cnt; ++j) {
                        const unsigned c = tile_base(s_words, g + j + 1, g_lds0);
                        if (!((a | b | c) & 12u)) {
                            const unsigned x = 16 * a + 4 * b + c;
                            if (SINGLE) ctx_single = x;
                            else s_hist[x][tid] += 1;
                        }
                        a = b;
                        b = c;
                    }
                }
            }
            for (int64_t c0 = 0; c0 < C; c0 += kTileCohorts) {
                const int nc = (int)(C - c0 < kTileCohorts ? C - c0 : kTileCohorts);
                __syncthreads();
                for (int i = tid; i < kTileCohorts * 64; i += kTileBlock) {
                    const int c = i >> 6, x = i & 63;
                    s_S[x][c] = c < nc ? s_prob[(c0 + c) * 64 + x] : 0.0;
                }
                __syncthreads();
                if (tid < nc) {
                    double T = 0.0;
                    for (int x = 0; x < 64; ++x) T = fma((double)s_H[x], s_S[x][tid], T);
                    s_T[tid] = T;
                }
                __syncthreads();
                if (SINGLE) {
                    for (int c = 0; c < nc; ++c) {
                        double v = nan;
                        if (live) v = (ctx_single < 64 ? s_S[ctx_single][c] : 0.0) / s_T[c];
                        if (t < n_tiles) __builtin_nontemporal_store(v, &pt[((c0 + c) * R + r) * n_tiles + t]);
                    }
                } else {
                    double acc[kTileCohorts];
#pragma unroll
                    for (int c = 0; c < kTileCohorts; ++c) acc[c] = 0.0;
                    if (live) {
                        for (int x = 0; x < 64; ++x) {
                            const unsigned h = s_hist[x][tid];
                            if (__any(h != 0)) {                 // (a wave of tiles without this context skips the row)
                                const double hv = (double)h;
                                const double2* row = reinterpret_cast<const double2*>(&s_S[x][0]);      // broadcast reads
#pragma unroll
                                for (int c = 0; c < kTileCohorts; c += 2) {
                                    const double2 sv = row[c >> 1];
                                    acc[c] = fma(hv, sv.x, acc[c]);
                                    acc[c + 1] = fma(hv, sv.y, acc[c + 1]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < kTileCohorts; ++c)
                        if (c < nc && t < n_tiles) __builtin_nontemporal_store(live ? acc[c] / s_T[c] : nan, &pt[((c0 + c) * R + r) * n_tiles + t]);
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void tile_mut_counts_kernel(const int32_t* __restrict__ pair_mut, const int32_t* __restrict__ pair_reg,
                                                              int64_t n_pairs, const int64_t* __restrict__ mut_start,
                                                              const int32_t* __restrict__ mut_cohort, const int64_t* __restrict__ first_pos,
                                                              const int32_t* __restrict__ n_valid, int binsize, int64_t n_tiles, int64_t R,
                                                              int32_t* __restrict__ k)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += stride) {
        const int64_t m = pair_mut[i], r = pair_reg[i];
        const int64_t off = mut_start[m] - first_pos[r];           // START must be one of the region's positions (:160-163)
        if (off < 0) continue;
        const int64_t t = off / binsize;
        if (t >= n_valid[r]) continue;
        atomicAdd(&k[((int64_t)mut_cohort[m] * R + r) * n_tiles + t], 1);
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_base_tile_probs(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                        int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R,
                        const double* s_prob, int64_t C, int binsize, int64_t n_tiles, double* pt, int64_t* first_pos,
                        int32_t* n_valid, void* stream)
{
    DIG_REQUIRE(R >= 0 && C >= 0 && n_words >= 2 && n_chrom >= 0 && n_tiles >= 0, "non-negative sizes, n_words >= 2 (pad words)");
    DIG_REQUIRE(binsize >= 1, "binsize >= 1");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && first_pos && n_valid,
                "non-null pointers");
    DIG_REQUIRE(C == 0 || n_tiles == 0 || (s_prob && pt), "s_prob and pt");
    DIG_REQUIRE(binsize <= (kTileMaxWords - 1) * 8, "binsize at most 12 280 positions");
    const int grid = grid_for(R * kTileBlock, kTileBlock, 4);
    if (binsize == 1)
        hipLaunchKernelGGL((base_tile_probs_kernel<true>), dim3(grid), dim3(kTileBlock), 0, (hipStream_t)stream, genome_words, n_words,
                           chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, binsize, n_tiles, pt, first_pos, n_valid);
    else
        hipLaunchKernelGGL((base_tile_probs_kernel<false>), dim3(grid), dim3(kTileBlock), 0, (hipStream_t)stream, genome_words, n_words,
                           chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, binsize, n_tiles, pt, first_pos, n_valid);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_tile_mut_counts(const int32_t* pair_mut, const int32_t* pair_reg, int64_t n_pairs, const int64_t* mut_start,
                        const int32_t* mut_cohort, const int64_t* first_pos, const int32_t* n_valid, int binsize, int64_t n_tiles,
                        int64_t R, int64_t C, int32_t* k, void* stream)
{
    DIG_REQUIRE(n_pairs >= 0 && binsize >= 1 && n_tiles >= 0 && R >= 0 && C >= 0, "non-negative sizes, binsize >= 1");
    const int64_t n = C * R * n_tiles;
    if (n == 0) return DIG_OK;
    DIG_REQUIRE(k && first_pos && n_valid, "non-null outputs / region tables");
    DIG_HIP_TRY(hipMemsetAsync(k, 0, (size_t)n * sizeof(int32_t), (hipStream_t)stream));
    if (n_pairs == 0) return DIG_OK;
    DIG_REQUIRE(pair_mut && pair_reg && mut_start && mut_cohort, "non-null pair / mutation arrays");
    hipLaunchKernelGGL(tile_mut_counts_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, (hipStream_t)stream, pair_mut, pair_reg,
                       n_pairs, mut_start, mut_cohort, first_pos, n_valid, binsize, n_tiles, R, k);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
