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
// Mapping: one wave per region, lane = four consecutive words (one 16-byte load, 32 centre positions) per step, the
// next step's loads issued before the current one is counted: a wave keeps 2 KB in flight, which is what the
// latency-bound word-per-lane form lacked (Little: 8 TB/s x ~1 us needs ~8 MB in flight chip-wide).  Per position the
// work is two VALU operations (a shift and an and-or) and one LDS add, plus ten operations per word:
//   * words strictly inside the region that hold no N take the fast path: the eight 4-bit codes are compacted to
//     2 bits each (three mask-and-fold steps per word), joined with the neighbouring bases into a 20-bit string, and
//     every 6-bit substring IS a histogram index (left base in the low bits: the output stage undoes that order);
//   * the loop covers the aligned 4-word groups that lie strictly inside the region; the up to four words before the
//     first and after the last such group (range checks) are taken by lanes 0..7 before the loop, groups holding an N
//     inside the loop, both through the per-position path;
//   * the histogram is private to (lane mod 16): hist[index][lane mod 16] -- no same-address serialisation on AAA / TTT
//     runs, at most four lanes per bank.  The copies are summed with a rotated read (lane = context reads copy
//     (j + lane) mod 16 at step j), with the reverse-complement bin for '-' strand regions.
// Measured on MI355X (288 000 windows of 10 kb, tools/probe/ctx_probe.hip): 0.70 ms = 2.05 TB/s of packed genome; the
// shared-histogram word-per-lane form took 1.37 ms.  What bounds it now is instruction issue (about 8 lane-operations
// per position all told): 16 or 32 copies, 5 to 10 resident workgroups per CU and removing the LDS update altogether
// all land within 10 %.
#include "dig_common.hpp"

namespace dig {

constexpr int kCtxBlock = 256;
#ifndef DIG_CTX_COPIES
#define DIG_CTX_COPIES 16
#endif
constexpr int kCtxCopies = DIG_CTX_COPIES;             // histogram copies per wave (lane mod kCtxCopies)
constexpr int kCtxRowShift = DIG_CTX_COPIES == 32 ? 7 : 6;   // log2 of the row size in bytes
static_assert(DIG_CTX_COPIES == 32 || DIG_CTX_COPIES == 16, "row shift");
#ifndef DIG_CTX_PER_CU
#define DIG_CTX_PER_CU 6
#endif

typedef __attribute__((address_space(3))) unsigned lds_u32;

#ifndef DIG_CTX_BUMP            // (tools/probe/ctx_probe.hip redefines the histogram update to price its parts)
// one ds_add_u32 at LDS byte address `addr`
#define DIG_CTX_BUMP(addr) \
    __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

__device__ __forceinline__ int revcomp_ctx64(int c)
{
    const int b0 = c >> 4, b1 = (c >> 2) & 3, b2 = c & 3;
    return ((3 - b2) << 4) | ((3 - b1) << 2) | (3 - b0);
}

// positions of genome word w (global bases 8 w .. 8 w + 7; prev / cur / next = words w - 1, w, w + 1) that lie in
// [gs, ge) and whose window holds no N
__device__ __forceinline__ void ctx_count_word_checked(uint32_t prev, uint32_t cur, uint32_t next, int64_t w, int64_t gs,
                                                       int64_t ge, unsigned lane_addr)
{
    const uint64_t x = (uint64_t)(prev >> 28) | ((uint64_t)cur << 4) | ((uint64_t)(next & 15u) << 36);
    const int64_t g0 = w << 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned tri = (unsigned)(x >> (4 * k)) & 0xfffu;   // nibbles: left, centre, right
        const int64_t g = g0 + k;
        if (g >= gs && g < ge && !(tri & 0xcccu)) {
            const unsigned idx = (tri & 3u) | (((tri >> 4) & 3u) << 2) | (((tri >> 8) & 3u) << 4);   // left in the low bits
            DIG_CTX_BUMP(idx * (4u * kCtxCopies) + lane_addr);
        }
    }
}

// an N-free word: 8 positions, no checks.  The eight 2-bit codes are folded together upwards (shift-left-or, mask:
// two operations per level), which leaves them at bits 14..29; the neighbouring bases go to bits 12..13 and 30..31, so
// base i of the 10-base window sits at bits 12 + 2 i and the byte offset of histogram row (6-bit substring k) is one
// shift and one and-or away: (z >> (6 + 2 k)) & 0xfc0 | lane_addr   (rows are 64 bytes; lane_addr = LDS address of the
// wave's size-aligned histogram + 4 (lane mod 16)).
__device__ __forceinline__ void ctx_count_word_fast(uint32_t prev, uint32_t cur, uint32_t next, unsigned lane_addr)
{
    uint32_t y = cur & 0x33333333u;
    y = ((y << 2) | y) & 0x3c3c3c3cu;        // 4-bit fields at bits 2..5 of every byte
    y = ((y << 4) | y) & 0x3fc03fc0u;        // 8-bit fields at bits 6..13 of every half
    y = ((y << 8) | y) & 0x3fffc000u;        // 16 bits at 14..29
    const uint32_t z = (((prev >> 16) & 0x3000u) | y) | (next << 30);
    const unsigned row_mask = 63u << kCtxRowShift;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned addr;          // (the compiler turns the or into and + add: three operations instead of two)
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(z >> (12 - kCtxRowShift + 2 * k)), "s"(row_mask), "v"(lane_addr));
        DIG_CTX_BUMP(addr);
    }
}

struct CtxGroup {
    uint4 v;              // array words 4 G .. 4 G + 3  (= genome words 4 G - 1 .. 4 G + 2: one leading pad word)
    uint32_t before, after;
};

__device__ __forceinline__ CtxGroup ctx_load_group(const uint32_t* __restrict__ words, int64_t n_words, int64_t G)
{
    CtxGroup g;
    const int64_t a = 4 * G;
    if (a + 4 < n_words) {
        g.v = *reinterpret_cast<const uint4*>(words + a);
        g.after = words[a + 4];
    } else {   // the last group of the array: clamp to the trailing pad word
        const int64_t last = n_words - 1;
        g.v = make_uint4(words[a < last ? a : last], words[a + 1 < last ? a + 1 : last], words[a + 2 < last ? a + 2 : last],
                         words[a + 3 < last ? a + 3 : last]);
        g.after = words[last];
    }
    g.before = words[a > 0 ? a - 1 : 0];
    return g;
}

#ifndef DIG_CTX_WAVES_PER_SIMD
#define DIG_CTX_WAVES_PER_SIMD 6
#endif
__global__ __launch_bounds__(kCtxBlock, DIG_CTX_WAVES_PER_SIMD) void context_count_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, const uint8_t* __restrict__ reg_minus, int64_t R, int32_t* __restrict__ out)
{
    __shared__ alignas(256 * kCtxCopies) uint4 hist_all[kCtxBlock / 64][64 * kCtxCopies / 4];     // 4 KB per wave, aligned to its size
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lane32 = lane & (kCtxCopies - 1);
    uint4* hist4 = hist_all[wave];
    unsigned* hist = reinterpret_cast<unsigned*>(hist4);
    // LDS byte address of this lane's histogram column
    const unsigned lane_addr = (unsigned)(uintptr_t)(lds_u32*)hist + 4u * lane32;
#pragma unroll
    for (int i = 0; i < 64 * kCtxCopies / 4 / 64; ++i) hist4[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    const int64_t wave0 = (int64_t)blockIdx.x * (kCtxBlock / 64) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (kCtxBlock / 64);
    for (int64_t r = wave0; r < R; r += nwaves) {
        const int ch = reg_chrom[r];
        const int minus = reg_minus[r];
        const int64_t len = chrom_len[ch], off = chrom_off[ch];
        int64_t s = reg_start[r], e = reg_end[r];
        if (s == 0) s = 1;                                  // fetch_sequence :25-26
        if (e > len - 1) e = len - 1;                       // the fetch is truncated: the last centre is len - 2
        // centre positions [s, e) of the chromosome = global bases [gs, ge); words carry a leading pad word
        const int64_t gs = off + s, ge = off + e;
        if (ge > gs) {
            const int64_t w0 = gs >> 3, w1 = (ge - 1) >> 3;
            // Aligned groups of four array words whose genome words 4 G - 1 .. 4 G + 2 all lie strictly inside (w0, w1):
            const int64_t Ga = ((w0 + 1) >> 2) + 1, Gb = ((w1 + 1) >> 2) - 1;
            // everything else -- at most four words before the first full group and four after the last -- is taken
            // by lanes 0..7 in one pass of the range-checked path
            if (lane < 8) {
                const int64_t head_end = w1 < 4 * Ga - 2 ? w1 : 4 * Ga - 2;
                const int64_t tail_start = 4 * (Gb + 1 > Ga ? Gb + 1 : Ga) - 1;
                const int64_t w = lane < 4 ? w0 + lane : tail_start + (lane - 4);
                if (lane < 4 ? w <= head_end : w <= w1)
                    ctx_count_word_checked(words[w], words[w + 1], words[w + 2], w, gs, ge, lane_addr);
            }
            int64_t G = Ga + lane;
            CtxGroup cur{};
            if (G <= Gb) cur = ctx_load_group(words, n_words, G);
            while (G <= Gb) {
                const int64_t Gn = G + 64;
                CtxGroup nxt{};
                if (Gn <= Gb) nxt = ctx_load_group(words, n_words, Gn);
                const uint32_t wd[6] = {cur.before, cur.v.x, cur.v.y, cur.v.z, cur.v.w, cur.after};
                const uint32_t any_n = ((cur.v.x | cur.v.y | cur.v.z | cur.v.w) & 0xccccccccu) | (cur.before & 0xc0000000u) |
                                       (cur.after & 0xcu);
                if (!any_n) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ctx_count_word_fast(wd[j], wd[j + 1], wd[j + 2], lane_addr);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ctx_count_word_checked(wd[j], wd[j + 1], wd[j + 2], 4 * G - 1 + j, gs, ge, lane_addr);
                }
                cur = nxt;
                G = Gn;
            }
        }
        // (one wave owns this histogram: LDS operations of a wave complete in order, no barrier needed)
        __builtin_amdgcn_wave_barrier();
        const int ctx = minus ? revcomp_ctx64(lane) : lane;                     // sequence_tools.py:527-566
        const int idx = ((ctx >> 4) & 3) | (ctx & 12) | ((ctx & 3) << 4);      // histogram index: left base in the low bits
        unsigned total = 0;
#pragma unroll
        for (int j = 0; j < kCtxCopies; ++j) total += hist[idx * kCtxCopies + ((j + lane) & (kCtxCopies - 1))];
        out[r * 64 + lane] = (int32_t)total;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 64 * kCtxCopies / 4 / 64; ++i) hist4[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_wave_barrier();
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
    DIG_REQUIRE(((uintptr_t)genome_words & 15) == 0, "genome_words 16-byte aligned");
    const int grid = grid_for(R * 64, kCtxBlock, DIG_CTX_PER_CU);
    hipLaunchKernelGGL(context_count_kernel, dim3(grid), dim3(kCtxBlock), 0, (hipStream_t)stream, genome_words, n_words,
                       chrom_off, chrom_len, reg_chrom, reg_start, reg_end, reg_minus, R, out);
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
