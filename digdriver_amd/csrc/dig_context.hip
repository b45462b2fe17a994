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


// =====================================================================================================================
// Round 4: the 2-bit form, dig_count_contexts2.
//
// The 4-bit kernel above needs 4.6 vector instructions per base (two per position for the histogram address, nine per word
// to squeeze the nibbles, the range / N checks), one LDS atomic per base -- an LDS atomic moves its address and its operand
// to the LDS at 4 cycles per wave-instruction: 16 bases per clock and CU, 293 us for a 2.88 Gb genome before any
// arithmetic -- and a 16-copy histogram per wave that is summed and cleared for every region.  Here
//   * the genome is resident at 2 BITS per base (A=0 C=1 G=2 T=3; every other letter stored as A) next to a sorted list
//     of the runs of non-ACGT letters: half the bytes, and the scan has no test for unknown letters at all;
//   * what is counted is the 4-MER at every SECOND base: the 4-mer at even base q holds the contexts of centres q + 1 and
//     q + 2, so one LDS atomic serves two positions (256 bins; the 64 context counts are two marginals of them);
//   * with four bases per byte the 4-mers at bases = 0 mod 4 ARE the bytes of a word and those at bases = 2 mod 4 the
//     bytes of the word shifted by four bits: one v_perm_b32 builds the LDS address {bin, column} of a 4-mer -- 9 vector
//     instructions and 8 atomics per 16 bases;
//   * ONE LANE PER REGION: a lane walks its own region and owns a histogram column of 256 sixteen-bit counters (lanes l and
//     l + 32 share the dwords of a column, low and high half: they sit in different half-waves and never meet in an LDS
//     cycle; the column index is the bank: no conflicts, nothing to sum, the column IS the region's histogram).  Two
//     waves share a 64 KB array (rows of 256 bytes = 2 waves x 32 dword columns), a workgroup of four waves holds two:
//     128 KB, one workgroup per CU -- the atomics' 16 lanes per clock are what bounds the kernel, not occupancy; every
//     lane keeps two 64-base groups in flight.  (The first two builds of this round gave a region to a wave, 16 columns
//     each: summing 16 KB of counters per region cost as much LDS time as counting a 10-kb region -- 0.79 ms, slower than
//     the 4-bit kernel.)
//   * the first and last group of a region are counted with a per-4-mer range test (wave-uniform branch), the one centre
//     at either end that no 4-mer covers is added as a context, centres whose window touches a non-ACGT run are
//     subtracted afterwards in the lane's own output row: a run's interior counts as AAA, its edge centres are looked up;
//   * a region of more than 131 068 bases is counted in segments (a counter holds 65 535).
// =====================================================================================================================
constexpr int kC2Block = 256;
constexpr int kC2PadBases = 64;            // bases in front of chromosome data (one 4-word group); >= 24 pad words behind (a lane reads whole 16-word steps)
constexpr int kC2BucketShift = 12;         // nint_bucket[b]: first run that ends behind base b << 12
constexpr int64_t kC2SegQuads = 65534;     // 4-mers per segment: no 16-bit counter can wrap

#ifndef DIG_C2_ABL
#define DIG_C2_ABL 0       // developer ablation builds (tools/build_variant.sh): 1 no LDS atomics (addresses still formed), 2 no global loads in the scan
#endif
#if DIG_C2_ABL & 1
#define DIG_C2_ADD(addr, val) asm volatile("" ::"v"(addr), "v"(val))
#else
#define DIG_C2_ADD(addr, val) \
    __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)), (unsigned)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

// the 6-bit code (left base in the low bits) of the window centred at base c
__device__ __forceinline__ unsigned c2_tri(const uint32_t* __restrict__ w, int64_t c)
{
    const int64_t q = c - 1;
    const uint64_t x = (uint64_t)w[q >> 4] | ((uint64_t)w[(q >> 4) + 1] << 32);
    return (unsigned)(x >> (2 * (int)(q & 15))) & 63u;
}

__device__ __forceinline__ int c2_code_to_ctx(unsigned m)            // code (left base low) -> context index 16 L + 4 C + R
{
    return (int)(((m & 3u) << 4) | (m & 12u) | (m >> 4));
}

struct C2Group {
    uint4 v;              // words 4 G .. 4 G + 3: bases 64 G .. 64 G + 63
    uint32_t after;       // word 4 G + 4 (its first base closes the group's last 4-mer)
};

__device__ __forceinline__ C2Group c2_load(const uint32_t* __restrict__ words, int64_t G)
{
    C2Group g;
    g.v = *reinterpret_cast<const uint4*>(words + 4 * G);
    g.after = words[4 * G + 4];
    return g;
}

struct C2Step {
    uint4 v[4];           // words 16 S .. 16 S + 15: bases 256 S .. 256 S + 255 (one 64-byte piece of the lane's stream)
    uint32_t after;       // word 16 S + 16
};

__device__ __forceinline__ C2Step c2_load_step(const uint32_t* __restrict__ words, int64_t S)
{
    C2Step t;
#if DIG_C2_ABL & 2
    for (int g = 0; g < 4; ++g) t.v[g] = make_uint4((uint32_t)S * 2654435761u, (uint32_t)S * 40503u, (uint32_t)S + g, (uint32_t)S ^ 0x9e3779b9u);
    t.after = (uint32_t)S;
    return t;
#endif
    const uint4* p = reinterpret_cast<const uint4*>(words + 16 * S);
#pragma unroll
    for (int g = 0; g < 4; ++g) t.v[g] = p[g];
    t.after = words[16 * S + 16];
    return t;
}

// all 32 even-base 4-mers of a group.  col_addr: LDS address of the lane's counter column in row 0 (dword aligned);
// inc: 1 or 1 << 16 (which half of the dword is the lane's counter).  A word's eight addresses are formed before its
// eight atomics are issued: an atomic that waits for its own address stalls the wave's whole instruction stream.
__device__ __forceinline__ void c2_count_group(const C2Group& g, unsigned col_addr, unsigned inc)
{
    const uint32_t w[5] = {g.v.x, g.v.y, g.v.z, g.v.w, g.after};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cur = w[j];
        const uint32_t sh = __builtin_amdgcn_alignbit(w[j + 1], cur, 4);      // bases 2 .. 17 of the word pair
        unsigned addr[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // {col_addr byte 3, col_addr byte 2, byte k of the source, col_addr byte 0}: a row (one 4-mer) is 256 bytes
            addr[2 * k] = __builtin_amdgcn_perm(cur, col_addr, 0x03020000u | ((4u + k) << 8));
            addr[2 * k + 1] = __builtin_amdgcn_perm(sh, col_addr, 0x03020000u | ((4u + k) << 8));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) DIG_C2_ADD(addr[k], inc);
        __builtin_amdgcn_sched_barrier(0);          // (without it the scheduler forms all 128 addresses of a step first and spills)
    }
}

// the same with a range test: 4-mer i of the group (base 64 G + 2 i) counts when lo <= i <= hi
__device__ __forceinline__ void c2_count_group_checked(const C2Group& g, unsigned col_addr, unsigned inc, int lo, int hi)
{
    const uint32_t w[5] = {g.v.x, g.v.y, g.v.z, g.v.w, g.after};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cur = w[j];
        const uint32_t sh = __builtin_amdgcn_alignbit(w[j + 1], cur, 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = 8 * j + 2 * k, i1 = i0 + 1;           // bases 16 j + 4 k and 16 j + 4 k + 2
            DIG_C2_ADD(__builtin_amdgcn_perm(cur, col_addr, 0x03020000u | ((4u + k) << 8)), (lo <= i0 && i0 <= hi) ? inc : 0u);
            DIG_C2_ADD(__builtin_amdgcn_perm(sh, col_addr, 0x03020000u | ((4u + k) << 8)), (lo <= i1 && i1 <= hi) ? inc : 0u);
        }
    }
}

// Layout: two 64 KB arrays per workgroup, one per pair of waves; row b (4-mer b) = 256 bytes = 2 waves x 32 dwords; lane l
// of a wave owns 16 bits of dword l mod 32 of its wave's half row (low half: lanes 0-31, high half: lanes 32-63).
__global__ __launch_bounds__(kC2Block, 1) void context_count2_kernel(
    const uint32_t* __restrict__ words, const int64_t* __restrict__ nint_start, const int64_t* __restrict__ nint_end, int64_t n_int,
    const int32_t* __restrict__ nint_bucket, int64_t n_buckets, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, const uint8_t* __restrict__ reg_minus, int64_t R, int32_t* __restrict__ out)
{
    __shared__ alignas(65536) uint32_t hist[2][256 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* half_row0 = reinterpret_cast<char*>(hist[wave >> 1]) + 128 * (wave & 1);        // the wave's 128 bytes of row 0
    // byte 1 of the address must be free for the 4-mer: both arrays start at multiples of 65 536
    const unsigned col_addr = (unsigned)(uintptr_t)(lds_u32*)half_row0 + 4u * (lane & 31);
    const unsigned inc = lane < 32 ? 1u : 0x10000u;
    const unsigned short* cnt = reinterpret_cast<const unsigned short*>(half_row0) + 2 * (lane & 31) + (lane >> 5);   // cnt[128 b]
    auto zero_rows = [&]() {                                     // 8 lanes clear the wave's 128 bytes of a row: 8 rows per instruction
#pragma unroll
        for (int t = 0; t < 32; ++t)
            *reinterpret_cast<uint4*>(half_row0 + 256 * (8 * t + (lane >> 3)) + 16 * (lane & 7)) = make_uint4(0u, 0u, 0u, 0u);
    };
    zero_rows();
    __builtin_amdgcn_wave_barrier();
    const int64_t n_lanes = (int64_t)gridDim.x * kC2Block;
    const int64_t wave_first = (int64_t)blockIdx.x * kC2Block + wave * 64;
    for (int64_t r0 = wave_first; r0 < R; r0 += n_lanes) {       // lane = region r0 + lane
        const int64_t r = r0 + lane;
        const bool have = r < R;
        int64_t gs = 0, ge = 0, jf = -1;
        int minus = 0;
        if (have) {
            const int ch = reg_chrom[r];
            minus = reg_minus[r];
            const int64_t len = chrom_len[ch], off = chrom_off[ch] + kC2PadBases;
            int64_t s = reg_start[r], e = reg_end[r];
            if (s == 0) s = 1;                                   // fetch_sequence :25-26
            if (e > len - 1) e = len - 1;                        // the fetch is truncated: the last centre is len - 2
            gs = off + s;
            ge = off + e;                                        // centres [gs, ge) in array bases
            if (ge < gs) ge = gs;
            if (n_int > 0 && ge > gs) {                          // first non-ACGT run that overlaps the widened region
                const int64_t x0 = gs - 1, x1 = ge + 1;
                int64_t b = x0 >> kC2BucketShift;
                if (b >= n_buckets) b = n_buckets - 1;
                int64_t jj = nint_bucket[b];
                while (jj < n_int && nint_end[jj] <= x0) ++jj;
                if (jj < n_int && nint_start[jj] < x1) jf = jj;
            }
        }
        const int64_t a = gs & ~(int64_t)1;                      // first even base >= gs - 1
        const int64_t n4 = ge > gs ? (ge - 1 - a) >> 1 : 0;      // 4-mers at a, a + 2, ...: centres a + 1 .. a + 2 n4
        // the centre in front of the first pair and the one behind the last
        int head_code = -1, tail_code = -1;
        if (ge > gs && a == gs) head_code = (int)c2_tri(words, gs);
        if (ge > gs && ((ge - 1 - a) & 1)) tail_code = (int)c2_tri(words, ge - 1);
        uint32_t T[64];                                          // context totals by CODE (left base in the low bits)
#pragma unroll
        for (int m = 0; m < 64; ++m) T[m] = 0u;
        int64_t done = 0;                                        // 4-mers counted so far
        for (;;) {
            const int64_t todo = n4 - done < kC2SegQuads ? n4 - done : kC2SegQuads;
            if (!__any(todo > 0)) break;
            if (todo > 0) {
                // 256-base steps (64 bytes of a lane's stream, four groups), two steps requested ahead of the one being counted
                const int64_t q0 = a + 2 * done, q1 = q0 + 2 * (todo - 1);
                const int64_t S0 = q0 >> 8, S1 = q1 >> 8;
                int64_t S = S0;
                C2Step cur = c2_load_step(words, S), nxt{}, nn{};
                if (S + 1 <= S1) nxt = c2_load_step(words, S + 1);
                while (S <= S1) {
                    if (S + 2 <= S1) nn = c2_load_step(words, S + 2);
                    if (__any(S == S0 || S == S1)) {
                        const int rel0 = (int)(q0 - 256 * S < -256 ? -256 : (q0 - 256 * S > 512 ? 512 : q0 - 256 * S));
                        const int rel1 = (int)(q1 - 256 * S < -256 ? -256 : (q1 - 256 * S > 512 ? 512 : q1 - 256 * S));
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int b0 = rel0 - 64 * g, b1 = rel1 - 64 * g;        // in bases, relative to the group
                            const int lo = b0 <= 0 ? 0 : (b0 >= 64 ? 32 : b0 >> 1);
                            const int hi = b1 < 0 ? -1 : (b1 >= 62 ? 31 : b1 >> 1);
                            C2Group grp;
                            grp.v = cur.v[g];
                            grp.after = g < 3 ? cur.v[g < 3 ? g + 1 : 3].x : cur.after;
                            c2_count_group_checked(grp, col_addr, inc, lo, hi);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            C2Group grp;
                            grp.v = cur.v[g];
                            grp.after = g < 3 ? cur.v[g < 3 ? g + 1 : 3].x : cur.after;
                            c2_count_group(grp, col_addr, inc);
                        }
                    }
                    cur = nxt;
                    nxt = nn;
                    ++S;
                }
            }
            done += todo;
            // ---- the lane's counters are this segment's histogram: add its two marginals to the totals, clear them ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                // a 4-mer b0 b1 b2 b3 (row b0 + 4 b1 + 16 b2 + 64 b3) holds the contexts (b0 b1 b2) and (b1 b2 b3)
                uint32_t acc = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc += (uint32_t)cnt[128 * (m + 64 * t)] + (uint32_t)cnt[128 * (4 * m + t)];
                T[m] += acc;
            }
            __builtin_amdgcn_wave_barrier();
            zero_rows();
            __builtin_amdgcn_wave_barrier();
        }
        if (have) {
#pragma unroll
            for (int m = 0; m < 64; ++m) T[m] += (uint32_t)(m == head_code) + (uint32_t)(m == tail_code);
            // out[r][ctx], ctx = 16 b0 + 4 b1 + b2; a '-' strand region reports the reverse complement
            // (sequence_tools.py:527-566): revcomp(ctx) = 63 - (16 b2 + 4 b1 + b0)
            int32_t* row = out + r * 64;
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {
                int32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ctx = 4 * c4 + i;
                    const int code_plus = ((ctx >> 4) & 3) | (ctx & 12) | ((ctx & 3) << 4);
                    const int rc = 63 - (((ctx & 3) << 4) | (ctx & 12) | ((ctx >> 4) & 3));
                    const int code_minus = ((rc >> 4) & 3) | (rc & 12) | ((rc & 3) << 4);
                    v[i] = (int32_t)(minus ? T[code_minus] : T[code_plus]);
                }
                *reinterpret_cast<int4*>(row + 4 * c4) = make_int4(v[0], v[1], v[2], v[3]);
            }
            // ---- centres whose window touches a non-ACGT run: taken back in the lane's own row ----
            if (jf >= 0) {
                const int64_t x1 = ge + 1;
                auto take = [&](unsigned code, int32_t n) {
                    int ctx = c2_code_to_ctx(code);
                    if (minus) ctx = revcomp_ctx64(ctx);
                    row[ctx] -= n;
                };
                for (int64_t j = jf; j < n_int; ++j) {
                    const int64_t ns = nint_start[j], ne = nint_end[j];
                    if (ns >= x1) break;                         // the list is sorted: nothing further can overlap
                    const int64_t prev_end = j > 0 ? nint_end[j - 1] : -1;      // centres up to prev_end belong to run j - 1
                    int64_t lo = ns - 1 > gs ? ns - 1 : gs;
                    if (prev_end + 1 > lo) lo = prev_end + 1;
                    const int64_t hi = ne + 1 < ge ? ne + 1 : ge;
                    if (hi <= lo) continue;
                    // interior: all three bases inside the run -> stored as AAA (code 0)
                    const int64_t i0 = lo > ns + 1 ? lo : ns + 1, i1 = hi < ne - 1 ? hi : ne - 1;
                    if (i1 > i0) take(0u, (int32_t)(i1 - i0));
                    const int64_t l1 = hi < ns + 1 ? hi : ns + 1;               // left edge centres [lo, l1)
                    int64_t e0 = ne - 1 > ns + 1 ? ne - 1 : ns + 1;             // right edge centres [e0, hi)
                    if (e0 < lo) e0 = lo;
                    for (int64_t c = lo; c < l1; ++c) take(c2_tri(words, c), 1);
                    for (int64_t c = e0; c < hi; ++c) take(c2_tri(words, c), 1);
                }
            }
        }
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

int dig_count_contexts2(const uint32_t* words2, int64_t n_words2, const int64_t* nint_start, const int64_t* nint_end, int64_t n_int,
                        const int32_t* nint_bucket, int64_t n_buckets, const int64_t* chrom_off, const int64_t* chrom_len, int n_chrom,
                        const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, const uint8_t* reg_minus, int64_t R,
                        int32_t* out, void* stream)
{
    DIG_REQUIRE(R >= 0 && n_words2 >= 28 && n_chrom >= 0 && n_int >= 0, "R, n_int, n_chrom >= 0, n_words2 >= 28 (pad words)");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(words2 && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && reg_minus && out, "non-null pointers");
    DIG_REQUIRE(n_int == 0 || (nint_start && nint_end && nint_bucket && n_buckets >= 1), "interval list with its bucket index");
    DIG_REQUIRE(((uintptr_t)words2 & 15) == 0, "words2 16-byte aligned");
    const int grid = grid_for(R, kC2Block, 1);          // one lane per region, one workgroup (128 KB of LDS) per CU
    hipLaunchKernelGGL(context_count2_kernel, dim3(grid), dim3(kC2Block), 0, (hipStream_t)stream, words2, nint_start, nint_end, n_int,
                       nint_bucket, n_buckets, chrom_off, chrom_len, reg_chrom, reg_start, reg_end, reg_minus, R, out);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

int dig_count_contexts2_host(const uint32_t* words2, int64_t n_words2, const int64_t* nint_start, const int64_t* nint_end, int64_t n_int,
                             const int32_t* nint_bucket, int64_t n_buckets, const int64_t* chrom_off, const int64_t* chrom_len,
                             int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end,
                             const uint8_t* reg_minus, int64_t R, int32_t* out, int device)
{
    DIG_REQUIRE(R >= 0 && n_words2 >= 28 && n_chrom >= 0 && n_int >= 0, "R, n_int, n_chrom >= 0, n_words2 >= 28 (pad words)");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(words2 && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && reg_minus && out, "non-null pointers");
    DIG_REQUIRE(n_int == 0 || (nint_start && nint_end && nint_bucket && n_buckets >= 1), "interval list with its bucket index");
    for (int64_t r = 0; r < R; ++r) {
        DIG_REQUIRE(reg_chrom[r] >= 0 && reg_chrom[r] < n_chrom, "region chromosome index within [0, n_chrom)");
        DIG_REQUIRE(reg_start[r] >= 0 && reg_end[r] >= 0, "non-negative coordinates");
    }
    for (int c = 0; c < n_chrom; ++c)
        DIG_REQUIRE(chrom_off[c] >= 0 && chrom_off[c] + chrom_len[c] + 64 <= (n_words2 - 24) * 16, "chromosomes inside the genome array");
    for (int64_t j = 0; j < n_int; ++j)
        DIG_REQUIRE(nint_start[j] < nint_end[j] && (j == 0 || nint_end[j - 1] < nint_start[j]), "intervals sorted, disjoint, not touching");
    DIG_HIP_TRY(hipSetDevice(device));
    DevBuf dw, dns, dne, dnb, doff, dlen, dc, ds, de, dm, dout;
#define UP(buf, src, bytes)        \
    DIG_HIP_TRY(buf.alloc(bytes)); \
    if ((bytes) > 0) DIG_HIP_TRY(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice))
    UP(dw, words2, (size_t)n_words2 * 4);
    UP(dns, nint_start, (size_t)n_int * 8);
    UP(dne, nint_end, (size_t)n_int * 8);
    UP(dnb, nint_bucket, (size_t)(n_int ? n_buckets : 0) * 4);
    UP(doff, chrom_off, (size_t)std::max(n_chrom, 1) * 8);
    UP(dlen, chrom_len, (size_t)std::max(n_chrom, 1) * 8);
    UP(dc, reg_chrom, (size_t)R * 4);
    UP(ds, reg_start, (size_t)R * 8);
    UP(de, reg_end, (size_t)R * 8);
    UP(dm, reg_minus, (size_t)R);
#undef UP
    DIG_HIP_TRY(dout.alloc((size_t)R * 64 * 4));
    int rc = dig_count_contexts2(dw.as<uint32_t>(), n_words2, dns.as<int64_t>(), dne.as<int64_t>(), n_int, dnb.as<int32_t>(), n_buckets,
                                 doff.as<int64_t>(), dlen.as<int64_t>(), n_chrom, dc.as<int32_t>(), ds.as<int64_t>(), de.as<int64_t>(),
                                 dm.as<uint8_t>(), R, dout.as<int32_t>(), nullptr);
    if (rc) return rc;
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpy(out, dout.p, (size_t)R * 64 * 4, hipMemcpyDeviceToHost));
    return DIG_OK;
}


}  // extern "C"
