// dig_context.hip -- trinucleotide context counting over an HBM-resident packed genome.
//
// Reference (one pysam fetch + a Python loop per region):
//   sequence_tools.py:21-29   fetch_sequence      region widened by one base on either side, START == 0 -> 1,
//                                                 truncated at the chromosome end, upper-cased
//   sequence_tools.py:42-55   seq_to_context      a window holding 'N' is skipped
//   sequence_tools.py:65-80   count_sequence_context   64 counts of the centre positions
//   sequence_tools.py:527-566 nonc_elt_context_count   '-' strand: the sequence is reverse-complemented first
// used by count_contexts_by_regions (:82-99, the 10-kb window counts), precount_region_contexts_parallel (:481-525,
// the element block counts L) and DIG_onthefly (driver_model/onthefly_tools.py:70-71,120).
//
// Genome layout: 4 bits per base (A=0, C=1, G=2, T=3, anything else = 4), eight bases per 32-bit word, base 0 in the
// low nibble; every chromosome starts on a word boundary and the array is padded with one all-N word at either end
// (hg19: 1.55 GB, one upload).  A pure streaming problem: 0.5 B per base.
//
// Mapping: one wave per region, lane = one word (8 centre positions) per step.  The lane builds a 40-bit window
// (last nibble of the previous word, its own word, first nibble of the next) and slides a 12-bit triplet over it;
// a triplet with any nibble >= 4 is skipped; the 64-bin histogram is per wave in LDS (ds_add_u32).  The result row is
// written with lane = context, reading the reverse-complement bin for '-' strand regions.
#include "dig_common.hpp"

namespace dig {

constexpr int kCtxBlock = 256;

__device__ __forceinline__ int revcomp_ctx64(int c)
{
    const int b0 = c >> 4, b1 = (c >> 2) & 3, b2 = c & 3;
    return ((3 - b2) << 4) | ((3 - b1) << 2) | (3 - b0);
}

__global__ __launch_bounds__(kCtxBlock) void context_count_kernel(
    const uint32_t* __restrict__ words, const int64_t* __restrict__ chrom_off, const int64_t* __restrict__ chrom_len,
    const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start, const int64_t* __restrict__ reg_end,
    const uint8_t* __restrict__ reg_minus, int64_t R, int32_t* __restrict__ out)
{
    __shared__ unsigned hist_all[kCtxBlock / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned* hist = hist_all[wave];
    const int64_t wave0 = (int64_t)blockIdx.x * (kCtxBlock / 64) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (kCtxBlock / 64);
    for (int64_t r = wave0; r < R; r += nwaves) {
        hist[lane] = 0;
        const int ch = reg_chrom[r];
        const int64_t len = chrom_len[ch], off = chrom_off[ch];
        int64_t s = reg_start[r], e = reg_end[r];
        if (s == 0) s = 1;                                  // fetch_sequence :25-26
        if (e > len - 1) e = len - 1;                       // the fetch is truncated: the last centre is len - 2
        // centre positions [s, e) of the chromosome = global bases [gs, ge); words carry a leading pad word
        const int64_t gs = off + s, ge = off + e;
        if (ge > gs) {
            const int64_t w0 = gs >> 3, w1 = (ge - 1) >> 3;
            for (int64_t w = w0 + lane; w <= w1; w += 64) {
                const uint32_t prev = words[w], cur = words[w + 1], next = words[w + 2];   // +1: leading pad word
                const uint64_t x = (uint64_t)(prev >> 28) | ((uint64_t)cur << 4) | ((uint64_t)(next & 15u) << 36);
                const int64_t g0 = w << 3;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned tri = (unsigned)(x >> (4 * k)) & 0xfffu;   // nibbles: left, centre, right
                    const int64_t g = g0 + k;
                    if (g >= gs && g < ge && !(tri & 0xcccu)) {
                        const unsigned ctx = ((tri & 3u) << 4) | (((tri >> 4) & 3u) << 2) | ((tri >> 8) & 3u);
                        atomicAdd(&hist[ctx], 1u);
                    }
                }
            }
        }
        // (one wave owns this histogram: LDS operations of a wave complete in order, no barrier needed)
        const int src = reg_minus[r] ? revcomp_ctx64(lane) : lane;
        out[r * 64 + lane] = (int32_t)hist[src];
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

int dig_count_contexts(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                       int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end,
                       const uint8_t* reg_minus, int64_t R, int32_t* out, void* stream)
{
    DIG_REQUIRE(R >= 0 && n_words >= 2 && n_chrom >= 0, "R >= 0, n_words >= 2 (pad words), n_chrom >= 0");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && reg_minus && out,
                "non-null pointers");
    const int grid = grid_for(R * 64, kCtxBlock, 8);
    hipLaunchKernelGGL(context_count_kernel, dim3(grid), dim3(kCtxBlock), 0, (hipStream_t)stream, genome_words, chrom_off,
                       chrom_len, reg_chrom, reg_start, reg_end, reg_minus, R, out);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_count_contexts_host(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off,
                            const int64_t* chrom_len, int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start,
                            const int64_t* reg_end, const uint8_t* reg_minus, int64_t R, int32_t* out, int device)
{
    DIG_REQUIRE(R >= 0 && n_words >= 2 && n_chrom >= 0, "R >= 0, n_words >= 2 (pad words), n_chrom >= 0");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && reg_minus && out,
                "non-null pointers");
    for (int64_t r = 0; r < R; ++r) {
        DIG_REQUIRE(reg_chrom[r] >= 0 && reg_chrom[r] < n_chrom, "region chromosome index within [0, n_chrom)");
        DIG_REQUIRE(reg_start[r] >= 0 && reg_end[r] >= 0, "non-negative coordinates");
    }
    for (int c = 0; c < n_chrom; ++c)
        DIG_REQUIRE((chrom_off[c] & 7) == 0 && chrom_off[c] + chrom_len[c] <= (n_words - 2) * 8,
                    "chromosomes word-aligned and inside the genome array");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dw, doff, dlen, dc, ds, de, dm, dout;
#define UP(buf, src, bytes)        \
    DIG_HIP_TRY(buf.alloc(bytes)); \
    DIG_HIP_TRY(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice))
    UP(dw, genome_words, (size_t)n_words * 4);
    UP(doff, chrom_off, (size_t)std::max(n_chrom, 1) * 8);
    UP(dlen, chrom_len, (size_t)std::max(n_chrom, 1) * 8);
    UP(dc, reg_chrom, (size_t)R * 4);
    UP(ds, reg_start, (size_t)R * 8);
    UP(de, reg_end, (size_t)R * 8);
    UP(dm, reg_minus, (size_t)R);
#undef UP
    DIG_HIP_TRY(dout.alloc((size_t)R * 64 * 4));
    int rc = dig_count_contexts(dw.as<uint32_t>(), n_words, doff.as<int64_t>(), dlen.as<int64_t>(), n_chrom,
                                dc.as<int32_t>(), ds.as<int64_t>(), de.as<int64_t>(), dm.as<uint8_t>(), R,
                                dout.as<int32_t>(), nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, (size_t)R * 64 * 4, hipMemcpyDeviceToHost));
    return DIG_OK;
}

}  // extern "C"
