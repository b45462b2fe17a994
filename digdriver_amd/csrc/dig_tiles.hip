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
#include <type_traits>

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

// Stage nw <= kTileMaxWords + 2 packed words (from array word w0 on, clamped to the trailing pad word) into LDS: all loads of a
// thread are issued before its first LDS write -- the plain loop `s_words[i] = words[...]` waits for every load in turn (one
// memory round trip per 256 words instead of one per region).
template <int BLOCK>
__device__ __forceinline__ void stage_words(uint32_t* s_words, const uint32_t* __restrict__ words, int64_t n_words, int64_t w0,
                                            int64_t nw, int tid)
{
    constexpr int kPer = (1538 + BLOCK - 1) / BLOCK;       // kTileMaxWords + 2
    uint32_t tmp[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int64_t i = tid + (int64_t)j * BLOCK;
        tmp[j] = i < nw ? words[(w0 + i < n_words ? w0 + i : n_words - 1)] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int64_t i = tid + (int64_t)j * BLOCK;
        if (i < nw) s_words[i] = tmp[j];
    }
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
            stage_words<256>(s_words, words, n_words, w0, nw, tid);
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
                stage_words<256>(s_words, words, n_words, w0, nw, tid);
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

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core form of the tile probabilities (binsize >= 2, at most 256 tiles and 12 280 tiled positions per region,
// cohorts in chunks of 48): pt[c][tile] = (sum_ctx S[c][ctx] h[ctx][tile]) / T[c] is a [C x 64] x [64 x tiles] FP64
// product per region.  v_mfma_f64_16x16x4_f64 has the FP64 vector rate on gfx950 and -- measured: the kernel's time is
// the SUM of its matrix and vector instruction time, whatever the occupancy -- shares the issue with the vector ALU,
// so the design minimises vector instructions around the 624 MFMAs of a region:
//   * A = S (16 cohorts x 4 contexts per instruction) lives in REGISTERS for the whole kernel (48 doubles per lane: the
//     table is the same for every region), lane 16 k + i holds S[c0 + 16 m + i][context of histogram row 4 ks + k]; the
//     rows are in the order the walk produces them (first base in the low bits: b0 + 4 b1 + 16 b2);
//   * B = h: lane 16 k + j reads h[4 ks + k][16 n + j] as one ds_read_u16 and converts it; the histogram rows are
//     272 counters apart so that the four k-rows of an operand fall into disjoint LDS banks;
//   * D[i][j] sits in lane 16 (i % 4) + j, register i / 4: sixteen consecutive tiles of one cohort per quarter wave --
//     128 contiguous bytes per non-temporal store;
//   * histograms: lane = tile walks its packed bases a WORD at a time: the eight 4-bit codes are squeezed to a 16-bit
//     string of 2-bit bases, joined to the two bases before them, and every context is one bit-field extract -- about
//     four instructions per base, one of them a ds_add_u32 without return (two 16-bit counters per dword).  A word with
//     a non-ACGT code, or one that straddles the tile's ends, takes the guarded form of the same routine (wave-uniform
//     choice);
//   * H = row sums of h, added as packed 16-bit pairs (no pair can overflow: a row sums to at most the 12 280 tiled
//     positions), + the positions behind the last tile, if any; T[c] from the A registers and H, per wave; the
//     quotient is a multiplication by 1 / T[c] (one division per cohort and region instead of one per tile).
// Work split: (cohort tile m, tile group n) pairs, n = (wave + m) mod 4 step 4 -- 10 / 9 / 10 / 10 pairs for
// 3 x 13.  Sum order differs from the vector form (groups of four contexts inside the MFMA): 1e-15 relative.
typedef double tile_double4 __attribute__((ext_vector_type(4)));
constexpr int kTmStride = 272;                 // 16-bit counters per histogram row (256 tiles + 16 of padding)
constexpr int kTmStride32 = kTmStride / 2;
constexpr int kTmChunk = 48;                   // cohorts per launch
#ifndef DIG_TM_OCC
#define DIG_TM_OCC 2                           // workgroups per CU (register budget 256)
#endif

// One packed word of a tile's walk: contexts centred on nibbles centre0 .. centre0 + 7 (centre0 = 8 word - 1).  GUARD: the
// word holds a non-ACGT base or straddles the tile's ends -- the centres that count are one 8-bit mask (inside [lo, hi), no
// non-ACGT base in the window), and a centre that does not count adds zero.
template <bool GUARD>
__device__ __forceinline__ void tile_word(unsigned w, unsigned& carry, unsigned& icarry, int centre0, int lo, int hi,
                                          unsigned* col, unsigned inc)
{
    unsigned x = w & 0x33333333u;                                  // 2-bit bases, squeezed: base n at bits 2 n
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0xFFFFu;
    const unsigned win = (x << 4) | carry;                         // the two bases before the word, then its eight
    carry = win >> 16;
    unsigned okm = 0xFFu;
    if (GUARD) {
        unsigned f = ((w >> 2) | (w >> 3)) & 0x11111111u;          // non-ACGT flags, squeezed: base n at bit n
        f = (f | (f >> 3)) & 0x03030303u;
        f = (f | (f >> 6)) & 0x000F000Fu;
        f = (f | (f >> 12)) & 0xFFu;
        const unsigned iwin = (f << 2) | icarry;                   // bit n + 2: base n of the word; bits n .. n + 2: the window of centre n
        icarry = iwin >> 8;
        const int a = lo - centre0, b = hi - centre0;              // centres a .. b - 1 of the word lie inside the tile
        const unsigned below_b = b >= 8 ? 0xFFu : (b <= 0 ? 0u : (1u << b) - 1u);
        const unsigned below_a = a >= 8 ? 0xFFu : (a <= 0 ? 0u : (1u << a) - 1u);
        okm = below_b & ~below_a & ~(iwin | (iwin >> 1) | (iwin >> 2));
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const unsigned ctx = (win >> (2 * n)) & 63u;
        const unsigned add = GUARD ? ((okm >> n) & 1u) * inc : inc;
        atomicAdd(col + ctx * kTmStride32, add);
    }
}

// The walk of one region: thread t owns tile t (positions t binsize .. of the n_cov tiled ones; lo0 = staged nibble
// coordinate of the region's first position) and counts its contexts into column t of the histogram.
__device__ __forceinline__ void tile_histograms(const uint32_t* s_words, uint32_t* s_hist32, int t, int nv, int binsize, int n_cov,
                                                int lo0)
{
    const bool live = t < nv;
    const int tp = t * binsize;
    int cnt = binsize;
    if (cnt > n_cov - tp) cnt = n_cov - tp;
    const int lo = live ? lo0 + tp : 0;                             // centres [lo, hi) in staged nibble coordinates
    const int hi = live ? lo + cnt : 0;
    const int w_last = live ? hi >> 3 : -1;                         // word of the right neighbour of the last position
    unsigned carry = 0, icarry = 0;
    const unsigned inc = 1u << (16 * (t & 1));
    unsigned* col = s_hist32 + (t >> 1);
    int wi = live ? (lo - 1) >> 3 : 0;
    unsigned w_next = s_words[wi];                                 // (a word ahead: its wait does not drain the atomics behind it)
    for (;; ++wi) {
        const bool go = wi <= w_last;
        if (!__any(go)) break;
        const unsigned w = go ? w_next : 0u;
        w_next = s_words[wi < kTileMaxWords ? wi + 1 : wi];
        const int centre0 = 8 * wi - 1;
        const bool plain = centre0 >= lo && centre0 + 7 < hi && (w & 0xCCCCCCCCu) == 0u && icarry == 0u;
        if (__all(plain || !go)) {                                 // (wave-uniform: the middle words of every tile)
            if (go) tile_word<false>(w, carry, icarry, centre0, lo, hi, col, inc);
        } else if (go) {
            tile_word<true>(w, carry, icarry, centre0, lo, hi, col, inc);
        }
    }
}

// 4-bit code of global base g straight from the packed array (array word = genome word + 1; clamped to the trailing pad word)
__device__ __forceinline__ unsigned tile_base_global(const uint32_t* __restrict__ words, int64_t n_words, int64_t g)
{
    const int64_t w = (g >> 3) + 1;
    return (words[w < n_words ? w : n_words - 1] >> (4 * (int)(g & 7))) & 15u;
}

#ifdef DIG_TM_TIMING                            // developer build: cycles per phase (wave 0 of every workgroup), tools/tile_variant_bench.py
__device__ unsigned long long g_tm_prof[8];
#define TM_MARK(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); tm_acc[k] += now_ - tm_last; tm_last = now_; } } while (0)
#else
#define TM_MARK(k) do {} while (0)
#endif

// MT full 16-cohort tiles (v_mfma_f64_16x16x4), then NQ quads of four cohorts (v_mfma_f64_4x4x4: its four 4x4 blocks take the
// SAME four cohort rows against four different groups of four tiles, so the B operand is the register of the 16-row form and a
// quad costs a quarter of a tile): 37 cohorts = 2 tiles + 2 quads pay for 2.5 tiles' worth of matrix time where three tiles
// paid for 3.  Same bits as whole tiles (measured: the two instructions sum their four products alike), so a cohort's values
// do not depend on the other cohorts of the call.  (A single cohort on the vector ALU -- sixteen FMAs per sixteen tiles --
// was 1 % faster still but rounds differently: not kept.)
template <int MT, int NQ = 0>
__global__ __launch_bounds__(kTileBlock, DIG_TM_OCC) void base_tile_probs_mfma_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, int64_t R, const double* __restrict__ s_prob, int64_t C, int c0, int binsize, int n_tiles,
    double* __restrict__ pt, int64_t* __restrict__ first_pos, int32_t* __restrict__ n_valid)
{
    __shared__ uint32_t s_words[kTileMaxWords + 2];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist32[64 * kTmStride32];
    __shared__ unsigned s_H[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const unsigned short* s_hist16 = reinterpret_cast<const unsigned short*>(s_hist32);

    constexpr int MTA = MT > 0 ? MT : 1, NQA = NQ > 0 ? NQ : 1;
    double A[MTA][16], Aq[NQA][16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int row = 4 * ks + lk;                                    // histogram row = b0 + 4 b1 + 16 b2 (walk order) ...
        const int ctx = ((row & 3) << 4) | (row & 12) | (row >> 4);     // ... of context 16 b0 + 4 b1 + b2
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t c = c0 + 16 * m + li;
            A[m][ks] = c < C ? s_prob[c * 64 + ctx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                                  // A[i][k] of every block b: lane 16 k + 4 b + i
            const int64_t c = c0 + 16 * MT + 4 * q + (lane & 3);
            Aq[q][ks] = c < C ? s_prob[c * 64 + ctx] : 0.0;
        }
    }
    const int n_groups = (n_tiles + 15) >> 4;
    const int64_t cohort_stride = R * n_tiles;                 // elements between two cohorts of pt
    const int64_t tiled = (int64_t)n_tiles * binsize;
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the A registers are in -- otherwise their first use inside the loop
                                               // waits on the counter, i.e. also for the stores of the previous region

#ifdef DIG_TM_TIMING
    unsigned long long tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm_last = __builtin_readcyclecounter();
#endif
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const TileRegion reg = tile_region(chrom_off, chrom_len, reg_chrom[r], reg_start[r], reg_end[r]);
        const int64_t tiles_valid = (reg.n_pos + binsize - 1) / binsize;
        const int nv = (int)(tiles_valid < n_tiles ? tiles_valid : n_tiles);
        const int n_cov = (int)(reg.n_pos < tiled ? reg.n_pos : tiled);        // positions that belong to a tile
        if (tid == 0) {
            first_pos[r] = reg.first;
            n_valid[r] = nv;
        }
        {
            uint4* z = reinterpret_cast<uint4*>(s_hist32);
            for (int i = tid; i < 64 * kTmStride32 / 4; i += kTileBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        const int64_t ga = reg.g0 - 1;                      // left neighbour of the first position
        const int64_t w0 = (ga >> 3) + 1;
        const int64_t g_lds0 = (w0 - 1) << 3;
        {
            const int nw = (int)(((ga + n_cov + 1) >> 3) + 1 - w0 + 1);
            stage_words<256>(s_words, words, n_words, w0, nw, tid);
        }
        TM_MARK(0);
        __syncthreads();
        TM_MARK(1);
        // ---- per-tile context histograms ----
        tile_histograms(s_words, s_hist32, tid, nv, binsize, n_cov, (int)(ga - g_lds0) + 1);
        TM_MARK(2);
        __syncthreads();
        TM_MARK(3);
        // ---- H[ctx] = sum over the tiles: lane = (row of the wave's sixteen, quarter of the row) ----
        {
            const int x = 16 * wave + li;
            const uint4* rowp = reinterpret_cast<const uint4*>(s_hist32 + x * kTmStride32 + 32 * lk);
            unsigned sum = 0;                                   // two 16-bit sums side by side
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 v = rowp[j];
                sum += (v.x + v.y) + (v.z + v.w);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (lane < 16) s_H[x] = (sum & 0xffffu) + (sum >> 16);
        }
        if (reg.n_pos > n_cov) {                              // positions behind the last tile count for the normalisation only
            __syncthreads();
            const int64_t pos_per_pass = (int64_t)(kTileMaxWords - 1) * 8;
            for (int64_t p0 = n_cov; p0 < reg.n_pos; p0 += pos_per_pass) {
                const int64_t np = reg.n_pos - p0 < pos_per_pass ? reg.n_pos - p0 : pos_per_pass;
                const int64_t gb = reg.g0 + p0 - 1;
                const int64_t wb0 = (gb >> 3) + 1;
                const int64_t nw = ((gb + np + 1) >> 3) + 1 - wb0 + 1;
                __syncthreads();
                stage_words<256>(s_words, words, n_words, wb0, nw, tid);
                __syncthreads();
                const int64_t gl = (wb0 - 1) << 3;
                for (int64_t j = tid; j < np; j += kTileBlock) {
                    const int64_t g = reg.g0 + p0 + j;
                    const unsigned a = tile_base(s_words, g - 1, gl), b = tile_base(s_words, g, gl), c = tile_base(s_words, g + 1, gl);
                    if (!((a | b | c) & 12u)) atomicAdd(&s_H[a + 4 * b + 16 * c], 1u);       // (row order of this kernel)
                }
            }
        }
        TM_MARK(4);
        __syncthreads();
        TM_MARK(5);
        // ---- 1 / T[c] (every wave for itself: no exchange), moved into the lane layout of D once per region ----
        double rt[MTA][4], rtq[NQA];
        {
            double t[MTA], tq[NQA];
#pragma unroll
            for (int m = 0; m < MT; ++m) t[m] = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) tq[q] = 0.0;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double hv = (double)s_H[4 * ks + lk];
#pragma unroll
                for (int m = 0; m < MT; ++m) t[m] = fma(hv, A[m][ks], t[m]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) tq[q] = fma(hv, Aq[q][ks], tq[q]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                t[m] += __shfl_xor(t[m], 16, 64);
                t[m] += __shfl_xor(t[m], 32, 64);
                const double inv = 1.0 / t[m];                 // lane i (any k) holds cohort 16 m + i
#pragma unroll
                for (int q = 0; q < 4; ++q) rt[m][q] = __shfl(inv, 4 * q + lk, 64);     // D register q: cohort 16 m + 4 q + k
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                tq[q] += __shfl_xor(tq[q], 16, 64);
                tq[q] += __shfl_xor(tq[q], 32, 64);
                rtq[q] = __shfl(1.0 / tq[q], lk, 64);          // lane i % 4 (any k, b) holds cohort i of the quad; D: cohort k
            }
        }
        // ---- the product ----
        double* const out_lane = pt + ((int64_t)(c0 + lk) * R + r) * n_tiles + li;       // cohort c0 + k, tile i
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            for (int n = (wave + m) & 3; n < n_groups; n += 4) {
                tile_double4 acc = {0.0, 0.0, 0.0, 0.0};
                const unsigned short* hp = s_hist16 + lk * kTmStride + 16 * n + li;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[m][ks], (double)hp[4 * ks * kTmStride], acc, 0, 0, 0);
                const int t = 16 * n + li;
                if (t < n_tiles) {
                    double* o = out_lane + 16 * n;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + 16 * m + 4 * q + lk < C)
                            __builtin_nontemporal_store(t < nv ? acc[q] * rt[m][q] : nan, o + (16 * m + 4 * q) * cohort_stride);
                }
            }
        }
        if constexpr (NQ > 0) {
            // the quads of a tile group share the converted histogram values
            for (int n = (wave + MT) & 3; n < n_groups; n += 4) {
                double accq[NQA];
#pragma unroll
                for (int q = 0; q < NQ; ++q) accq[q] = 0.0;
                const unsigned short* hp = s_hist16 + lk * kTmStride + 16 * n + li;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const double b = (double)hp[4 * ks * kTmStride];       // B[k][j] of block b': lane 16 k + 4 b' + j = the 16-column register
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accq[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aq[q][ks], b, accq[q], 0, 0, 0);
                }
                const int t = 16 * n + li;
                if (t < n_tiles) {
                    double* o = out_lane + 16 * n;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)                            // D[i][j] of block b': lane 16 i + 4 b' + j -- cohort k, tile i of this lane
                        if (c0 + 16 * MT + 4 * q + lk < C)
                            __builtin_nontemporal_store(t < nv ? accq[q] * rtq[q] : nan, o + (16 * MT + 4 * q) * cohort_stride);
                }
            }
        }
        TM_MARK(6);
        __syncthreads();
        TM_MARK(7);
    }
#ifdef DIG_TM_TIMING
    if (tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_tm_prof[k], tm_acc[k]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Two-role form of the matrix kernel (round 5; what dig_base_tile_probs launches).  tools/probe/mfma_f64_mix.hip: a chain of
// DEPENDENT v_mfma_f64_16x16x4 issues every 64 clocks (the pipe's rate) whatever a second wave of the SIMD does, an FP64 vector
// instruction of that second wave waits for a whole matrix instruction (73 clocks each) and an integer one gets about every
// second slot.  The kernel above ran its matrix instructions 130 - 160 clocks apart because every one of them waited for the
// v_cvt_f64_u32 of its B operand (an FP64 vector instruction behind the previous matrix instruction), and all four waves of a
// workgroup sat in the same phase: global-load latency, LDS atomics and barriers were nobody's matrix time.  Here:
//   * one workgroup of EIGHT waves per CU, two roles: waves 4-7 ("walkers", one per SIMD) build the histograms of region i + 1
//     while waves 0-3 ("multipliers", one per SIMD) run the product of region i out of the other histogram buffer: ONE barrier
//     per region, the walkers' latencies (packed words two regions ahead, region descriptions three ahead, 10 000 LDS atomics)
//     are filled by the multipliers' matrix instructions;
//   * no FP64 vector instruction in the product: a count becomes a double with three integer instructions (the high word of
//     the double of n < 2^20 is the float's exponent and mantissa moved by three bits; the low word is zero), once per
//     (histogram row group, tile group) for ALL cohort tiles and quads of the chunk (up to four independent chains);
//   * no barrier inside a role: a walker wave zeroes, fills and sums the histogram columns of ITS 64 tiles (LDS operations of a
//     wave execute in order), the region histogram H arrives as four partial rows per context; every multiplier wave forms
//     T and 1 / T for itself;
//   * whole tile groups go round the four multipliers, the cohort tiles / the quads of the last G mod 4 groups one by one
//     (13 groups x (2 tiles + 2 quads): 8.5 / 8.5 / 8 / 7.5 tile-times instead of 10 / 7.5 / 7.5 / 7.5).
// Same bits as the kernel above (same chains, same order; tests/test_gpu_tiles.py compares the two forms).
constexpr int kTsBlock = 512;
constexpr int kTsPer = (kTileMaxWords + 2 + 255) / 256;         // packed words a walker thread carries for the next region

__device__ __forceinline__ double count_as_double(unsigned n)   // n < 2^20, exact, integer instructions only
{
    const unsigned f = __float_as_uint((float)n);
    const unsigned hi = n ? (f >> 3) + 0x38000000u : 0u;        // exponent 127 -> 1023; the 20 mantissa bits in use fit the high word
    return __hiloint2double((int)hi, 0);
}

__device__ __forceinline__ TileRegion tile_region_of(int64_t len, int64_t off, int64_t start, int64_t end)
{
    TileRegion t;
    t.first = start == 0 ? 1 : start;
    const int64_t stop = end < len - 1 ? end : len - 1;
    t.n_pos = stop > t.first ? stop - t.first : 0;
    t.g0 = off + t.first;
    return t;
}

// One tile group: the chains named by MASK (bit m = cohort tile m, bit MT = the quads) share the converted histogram values.
// The counts travel two steps ahead of the matrix instructions that take them (read at ks - 2, converted at ks - 1 behind the
// issue of step ks - 1's matrix instructions, i.e. while the last of them runs): written as it stands, the sequence
// read -> wait -> convert -> multiply left the matrix pipe idle for the LDS round trip and five dependent vector instructions
// in every step.  The plane stride reaches the stores through an opaque scalar so that the ten plane addresses are formed at the
// store (one 64-bit add each) instead of living in twenty registers across the loop.
template <int MT, int NQ, unsigned MASK>
__device__ __forceinline__ void ts_group(const double (&A)[MT > 0 ? MT : 1][16], const double (&Aq)[NQ > 0 ? NQ : 1][16],
                                         const unsigned short* hp, double* o, bool col_ok, bool tile_ok,
                                         const double (&rt)[MT > 0 ? MT : 1][4], const double (&rtq)[NQ > 0 ? NQ : 1], int c0, int64_t C,
                                         int lk, int64_t cohort_stride)
{
    constexpr int MTA = MT > 0 ? MT : 1, NQA = NQ > 0 ? NQ : 1;
    constexpr bool QUADS = NQ > 0 && ((MASK >> MT) & 1u);
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    tile_double4 acc[MTA];
    double accq[NQA];
#pragma unroll
    for (int m = 0; m < MTA; ++m) acc[m] = tile_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQA; ++q) accq[q] = 0.0;
    unsigned raw = hp[0];
    double b = count_as_double(raw);
    raw = hp[4 * kTmStride];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if ((MASK >> m) & 1u) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[m][ks], b, acc[m], 0, 0, 0);
        if (QUADS) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) accq[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aq[q][ks], b, accq[q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks < 15) b = count_as_double(raw);
        if (ks < 14) raw = hp[4 * (ks + 2) * kTmStride];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (col_ok) {
        int64_t cs = cohort_stride;
        asm volatile("" : "+s"(cs));
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if ((MASK >> m) & 1u) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + 16 * m + 4 * q + lk < C)
                        __builtin_nontemporal_store(tile_ok ? acc[m][q] * rt[m][q] : nan, o + (16 * m + 4 * q) * cs);
            }
        if (QUADS) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (c0 + 16 * MT + 4 * q + lk < C)
                    __builtin_nontemporal_store(tile_ok ? accq[q] * rtq[q] : nan, o + (16 * MT + 4 * q) * cs);
        }
    }
}

template <int MT, int NQ = 0>
__global__ __launch_bounds__(kTsBlock, 1) void base_tile_probs_roles_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, int64_t R, const double* __restrict__ s_prob, int64_t C, int c0, int binsize, int n_tiles,
    double* __restrict__ pt, int64_t* __restrict__ first_pos, int32_t* __restrict__ n_valid)
{
    __shared__ uint32_t s_words[2][kTileMaxWords + 2];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist32[2][64 * kTmStride32];
    __shared__ __attribute__((aligned(16))) unsigned s_Hp[2][64][4];      // region histogram: one partial per walker wave
    __shared__ int s_nv[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool walker = wave >= 4;
    const int li = lane & 15, lk = lane >> 4;
    constexpr int MTA = MT > 0 ? MT : 1, NQA = NQ > 0 ? NQ : 1;
    constexpr int P = MT + (NQ > 0 ? 1 : 0);                               // chains of a tile group that can be dealt singly
    constexpr unsigned ALL = (1u << P) - 1u;
    const int64_t n_my = blockIdx.x < R ? (R - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int n_groups = (n_tiles + 15) >> 4;
    const int64_t cohort_stride = R * n_tiles;
    const int64_t tiled = (int64_t)n_tiles * binsize;

    double A[MTA][16], Aq[NQA][16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int row = 4 * ks + lk;                                    // histogram row = b0 + 4 b1 + 16 b2 (walk order) ...
        const int ctx = ((row & 3) << 4) | (row & 12) | (row >> 4);     // ... of context 16 b0 + 4 b1 + b2
#pragma unroll
        for (int m = 0; m < MTA; ++m) {
            const int64_t c = c0 + 16 * m + li;
            A[m][ks] = (!walker && m < MT && c < C) ? s_prob[c * 64 + ctx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int64_t c = c0 + 16 * MT + 4 * q + (lane & 3);
            Aq[q][ks] = (!walker && q < NQ && c < C) ? s_prob[c * 64 + ctx] : 0.0;
        }
    }

    // ---- walker state: region i (d_cur), i + 1 (d_nxt: its words travel during the walk of i), i + 2 (raw: chromosome, start, end) ----
    const int ht = tid - 256, hw = wave - 4;
    auto region_index = [&](int64_t i) { const int64_t r = blockIdx.x + i * gridDim.x; return r < R ? r : R - 1; };
    TileRegion d_cur = {0, 0, 0}, d_nxt = {0, 0, 0};
    int raw_chrom = 0;
    int64_t raw_start = 0, raw_end = 0;
    auto words_of = [&](const TileRegion& d, int64_t& w0, int& nw, int& n_cov) {
        n_cov = (int)(d.n_pos < tiled ? d.n_pos : tiled);
        const int64_t ga = d.g0 - 1;
        w0 = (ga >> 3) + 1;
        nw = (int)(((ga + n_cov + 1) >> 3) + 1 - w0 + 1);
    };
    if (walker && n_my > 0) {
        const int64_t r0 = region_index(0), r1 = region_index(1), r2 = region_index(2);
        const int ch0 = reg_chrom[r0], ch1 = reg_chrom[r1];
        d_cur = tile_region_of(chrom_len[ch0], chrom_off[ch0], reg_start[r0], reg_end[r0]);
        d_nxt = tile_region_of(chrom_len[ch1], chrom_off[ch1], reg_start[r1], reg_end[r1]);
        raw_chrom = reg_chrom[r2];
        raw_start = reg_start[r2];
        raw_end = reg_end[r2];
        int64_t w0;
        int nw, n_cov;
        words_of(d_cur, w0, nw, n_cov);
        stage_words<256>(s_words[0], words, n_words, w0, nw, ht);
    }
    __syncthreads();

#ifdef DIG_TM_TIMING                            // walker leader: 0 loads issued | 1 zero | 2 walk | 3 sums | 4 words + description | 5 barrier;
    unsigned long long tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm_last = __builtin_readcyclecounter();      // multiplier leader: 6 product | 7 barrier
#define TS_MARK(k) do { if (lane == 0 && (wave == 0 || wave == 4)) { const unsigned long long now_ = __builtin_readcyclecounter(); tm_acc[k] += now_ - tm_last; tm_last = now_; } } while (0)
#else
#define TS_MARK(k) do {} while (0)
#endif
    for (int64_t i = 0; i <= n_my; ++i) {
        const int b = (int)(i & 1);
        if (walker) {
            if (i < n_my) {
                const int64_t r = blockIdx.x + i * gridDim.x;
                // the next region's packed words: loads now, LDS writes behind the walk
                uint32_t wreg[kTsPer];
                int64_t w0n;
                int nwn, n_cov_n;
                words_of(d_nxt, w0n, nwn, n_cov_n);
                if (i + 1 >= n_my) nwn = 0;
#pragma unroll
                for (int j = 0; j < kTsPer; ++j) {
                    const int64_t k = ht + (int64_t)j * 256;
                    wreg[j] = k < nwn ? __builtin_nontemporal_load(&words[(w0n + k < n_words ? w0n + k : n_words - 1)]) : 0u;
                }
                // the description of region i + 2 (its chromosome's row) and the raw row of region i + 3
                const int64_t len2 = chrom_len[raw_chrom], off2 = chrom_off[raw_chrom];
                const int64_t r3 = region_index(i + 3);
                const int ch3 = reg_chrom[r3];
                const int64_t st3 = reg_start[r3], en3 = reg_end[r3];

                const TileRegion reg = d_cur;
                const int64_t tiles_valid = (reg.n_pos + binsize - 1) / binsize;
                const int nv = (int)(tiles_valid < n_tiles ? tiles_valid : n_tiles);
                const int n_cov = (int)(reg.n_pos < tiled ? reg.n_pos : tiled);
                if (ht == 0) {
                    first_pos[r] = reg.first;
                    n_valid[r] = nv;
                    s_nv[b] = nv;
                }
                uint32_t* hist = s_hist32[b];
                TS_MARK(0);
                {   // zero the columns of this wave's tiles (32 dwords of every row): eight rows per 16-byte store
                    uint4* z = reinterpret_cast<uint4*>(hist + 32 * hw + 4 * (lane & 7)) ;
#pragma unroll
                    for (int k = 0; k < 8; ++k) z[((lane >> 3) + 8 * k) * (kTmStride32 / 4)] = make_uint4(0u, 0u, 0u, 0u);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                TS_MARK(1);
                const int64_t ga = reg.g0 - 1;
                const int64_t g_lds0 = (((ga >> 3) + 1) - 1) << 3;
                tile_histograms(s_words[b], hist, ht, nv, binsize, n_cov, (int)(ga - g_lds0) + 1);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                TS_MARK(2);
                {   // this wave's share of H: lane = histogram row
                    const uint4* rowp = reinterpret_cast<const uint4*>(hist + lane * kTmStride32 + 32 * hw);
                    unsigned sum = 0;                                   // two 16-bit sums side by side
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint4 v = rowp[j];
                        sum += (v.x + v.y) + (v.z + v.w);
                    }
                    s_Hp[b][lane][hw] = (sum & 0xffffu) + (sum >> 16);
                }
                if (reg.n_pos > n_cov) {                              // positions behind the last tile count for the normalisation only
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    for (int64_t j = n_cov + ht; j < reg.n_pos; j += 256) {
                        const int64_t g = reg.g0 + j;
                        const unsigned a = tile_base_global(words, n_words, g - 1), bb = tile_base_global(words, n_words, g),
                                       c = tile_base_global(words, n_words, g + 1);
                        if (!((a | bb | c) & 12u)) atomicAdd(&s_Hp[b][a + 4 * bb + 16 * c][hw], 1u);
                    }
                }
                TS_MARK(3);
                // the next region's words (the buffer was last read in iteration i - 1)
#pragma unroll
                for (int j = 0; j < kTsPer; ++j) {
                    const int k = ht + j * 256;
                    if (k < nwn) s_words[b ^ 1][k] = wreg[j];
                }
                d_cur = d_nxt;
                d_nxt = tile_region_of(len2, off2, raw_start, raw_end);
                raw_chrom = ch3;
                raw_start = st3;
                raw_end = en3;
                TS_MARK(4);
            }
        } else if (i > 0) {
            const int bp = b ^ 1;
            const int64_t r = blockIdx.x + (i - 1) * gridDim.x;
            const int nv = s_nv[bp];
            // ---- 1 / T[c] (every wave for itself), moved into the lane layout of D ----
            double rt[MTA][4], rtq[NQA];
            {
                double t[MTA], tq[NQA];
#pragma unroll
                for (int m = 0; m < MTA; ++m) t[m] = 0.0;
#pragma unroll
                for (int q = 0; q < NQA; ++q) tq[q] = 0.0;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const uint4 h4 = *reinterpret_cast<const uint4*>(&s_Hp[bp][4 * ks + lk][0]);
                    const double hv = (double)((h4.x + h4.y) + (h4.z + h4.w));
#pragma unroll
                    for (int m = 0; m < MT; ++m) t[m] = fma(hv, A[m][ks], t[m]);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) tq[q] = fma(hv, Aq[q][ks], tq[q]);
                }
#pragma unroll
                for (int m = 0; m < MTA; ++m) {
                    t[m] += __shfl_xor(t[m], 16, 64);
                    t[m] += __shfl_xor(t[m], 32, 64);
                    const double inv = 1.0 / t[m];                 // lane i (any k) holds cohort 16 m + i
#pragma unroll
                    for (int q = 0; q < 4; ++q) rt[m][q] = __shfl(inv, 4 * q + lk, 64);     // D register q: cohort 16 m + 4 q + k
                }
#pragma unroll
                for (int q = 0; q < NQA; ++q) {
                    tq[q] += __shfl_xor(tq[q], 16, 64);
                    tq[q] += __shfl_xor(tq[q], 32, 64);
                    rtq[q] = __shfl(1.0 / tq[q], lk, 64);          // lane i % 4 (any k, b) holds cohort i of the quad; D: cohort k
                }
            }
            // ---- the product ----
            const unsigned short* s_hist16 = reinterpret_cast<const unsigned short*>(s_hist32[bp]);
            double* const out_lane = pt + ((int64_t)(c0 + lk) * R + r) * n_tiles + li;       // cohort c0 + k, tile i
            const int g_full = n_groups & ~3;
            for (int n = wave; n < g_full; n += 4) {
                const int t = 16 * n + li;
                ts_group<MT, NQ, ALL>(A, Aq, s_hist16 + lk * kTmStride + 16 * n + li, out_lane + 16 * n, t < n_tiles, t < nv, rt, rtq, c0,
                                      C, lk, cohort_stride);
            }
            for (int u = wave; u < (n_groups - g_full) * P; u += 4) {
                const int n = g_full + u / P, part = u % P;
                const int t = 16 * n + li;
                const unsigned short* hp = s_hist16 + lk * kTmStride + 16 * n + li;
                double* o = out_lane + 16 * n;
                if (part == 0) ts_group<MT, NQ, 1u>(A, Aq, hp, o, t < n_tiles, t < nv, rt, rtq, c0, C, lk, cohort_stride);
                if constexpr (P > 1)
                    if (part == 1) ts_group<MT, NQ, 2u>(A, Aq, hp, o, t < n_tiles, t < nv, rt, rtq, c0, C, lk, cohort_stride);
                if constexpr (P > 2)
                    if (part == 2) ts_group<MT, NQ, 4u>(A, Aq, hp, o, t < n_tiles, t < nv, rt, rtq, c0, C, lk, cohort_stride);
                if constexpr (P > 3)
                    if (part == 3) ts_group<MT, NQ, 8u>(A, Aq, hp, o, t < n_tiles, t < nv, rt, rtq, c0, C, lk, cohort_stride);
            }
            TS_MARK(6);
        }
        __syncthreads();
        if (walker) TS_MARK(5);
        else TS_MARK(7);
    }
#ifdef DIG_TM_TIMING
    if (lane == 0 && (wave == 0 || wave == 4))
        for (int k = 0; k < 8; ++k) atomicAdd(&g_tm_prof[k], tm_acc[k]);
#endif
}

// =======================================================================================
// General-context form: n_up = n_down = U, U = 2 is the DEFAULT signature of the reference's per-base functions
// (penta-nucleotide contexts: base_probabilities_by_region / apply_nb_to_region / nb_model, sequence_tools.py:292,
// nb_model.py:126,188); U = 1 reproduces the trinucleotide kernels above (used to cross-check this one).
//   region r: positions first .. stop - 1 with first = (start == 0 ? U : start), stop = min(end, chrom_len - U)
//       (fetch_sequence :21-29: START == 0 becomes n_up, the fetch is widened by U bases on either side and cut at the
//       chromosome end); a position whose (2 U + 1)-base window holds a non-ACGT base has probability 0.
// 4^(2U+1) contexts (1 024 for U = 2) rule out the per-tile histograms and the 64-row matrix product of the trinucleotide
// kernels: pt[c][tile] = sum over the tile's positions of S[c][context(position)] is a gather-sum, 13.3 G table reads per
// 36 000 bins x 37 cohorts.  Round 3's kernel (a lane = a tile, table [cohort][context] of 16 cohorts = 128 KB, one
// workgroup of four waves per CU, 64 lanes reading 64 random 8-byte entries: 5-6 lanes deep on a bank pair, every lane
// walking its own bases and contexts) took 13.9 ms per 36 000 bins.  Round 4:
//   * the table of a pass is [context][8 cohorts]: a row is 64 contiguous bytes, EIGHT LANES (the 8 cohorts) walk a tile
//     together -- one ds_read_b64 per wave and 8 positions, each group reading one row; 64 KB per pass, five passes for 37
//     cohorts, so that a workgroup of EIGHT waves (two per SIMD) fits a CU with its buffers;
//   * the contexts of a region are formed ONCE per pass by all 512 threads (one position per thread and step, two words
//     of the packed genome each) and staged in LDS as 16-bit codes -- left base in the low bits, the table is staged in
//     that order -- with code 4^(2U+1) = an all-zero row for a window that holds a non-ACGT base: the walk has no test;
//   * tile sums wait in LDS ([cohort][tile], stride 513: the eight writers of a tile hit eight banks) until the region
//     total is known, then leave as pt = sum / total with a lane = a tile (coalesced 8-byte stores per cohort plane);
//     a region with more than 512 tiles (binsize 1) is walked twice: totals first, then 512 tiles at a time.
//   pt = (sum over the tile) / (sum over the region): the reference normalises every position first and then sums (a few
//   ulp apart, as for the trinucleotide kernels).  Regions of up to 16 384 positions; a longer one is NOT evaluated: its
//   n_valid is -1 and its pt NaN (the host-side callers refuse such regions before the launch: engine.check_tile_regions).
// =======================================================================================
constexpr int kCtxBlock = 1024;           // threads per workgroup: kCtxWalkers walkers of 8 lanes (512: 5.5 ms where 1024 take 4.6)
constexpr int kCtxWalkers = kCtxBlock / 8;
constexpr int kCtxCoh = 8;                // cohorts per pass = lanes per walker
constexpr int kCtxMaxPos = 16384;         // positions of a region whose codes are staged
constexpr int kCtxSumTiles = 512;         // tile sums kept per walk
constexpr int kCtxSumStride = kCtxSumTiles + 1;

struct CtxRaw {                           // a region as handed in
    int chrom;
    int64_t start, end;
};
struct CtxRegion {                        // wave-uniform description of a region
    int64_t first, n_pos, g0, tiles_valid, w0, nw;
    bool too_long;
};

template <int U>
__device__ __forceinline__ CtxRegion ctx_region(const CtxRaw& a, int64_t len, int64_t off, int binsize)
{
    CtxRegion q;
    q.first = a.start == 0 ? U : a.start;
    const int64_t stop = a.end < len - U ? a.end : len - U;
    q.n_pos = stop > q.first ? stop - q.first : 0;
    q.g0 = off + q.first;
    q.tiles_valid = (q.n_pos + binsize - 1) / binsize;
    q.too_long = q.n_pos > kCtxMaxPos;
    const int64_t ga0 = q.g0 - U;                     // leftmost base of the first window
    q.w0 = (ga0 >> 3) + 1;                            // array word = genome word + 1 (leading pad word)
    q.nw = (q.n_pos > 0 && !q.too_long) ? ((ga0 + q.n_pos + 2 * U - 1) >> 3) + 1 - q.w0 + 3 : 0;
    return q;
}

constexpr int kCtxWordsPer = (kCtxMaxPos / 8 + 8 + kCtxBlock - 1) / kCtxBlock;       // packed words a thread stages per region
constexpr int kCtxSplitSlots = kCtxWalkers;       // extra unit sums: the pieces of the last tiles of a region (below)

template <int U>
__global__ __launch_bounds__(kCtxBlock) void base_tile_probs_ctx_kernel(
    const uint32_t* __restrict__ words, int64_t n_words, const int64_t* __restrict__ chrom_off,
    const int64_t* __restrict__ chrom_len, const int32_t* __restrict__ reg_chrom, const int64_t* __restrict__ reg_start,
    const int64_t* __restrict__ reg_end, int64_t R, const double* __restrict__ s_prob, int64_t C, int binsize, int64_t n_tiles,
    double* __restrict__ pt, int64_t* __restrict__ first_pos, int32_t* __restrict__ n_valid, int only_deferred)
{
    constexpr int W = 2 * U + 1;                      // window
    constexpr int K = 1 << (2 * W);                   // contexts
    __shared__ double s_S[K + 1][kCtxCoh];            // row K: zeros (a window with a non-ACGT base)
    __shared__ alignas(16) unsigned short s_code[kCtxMaxPos + 16];       // (+ 16: a trip of the walk reads sixteen codes from any p < n_pos - 3)
    __shared__ uint32_t s_words[kCtxMaxPos / 8 + 8];
    __shared__ double s_sum[kCtxCoh][kCtxSumStride + kCtxSplitSlots];      // tile sums, then the pieces of split tiles
    __shared__ double s_part[kCtxBlock / 8][kCtxCoh];
    __shared__ double s_T[kCtxWalkers / 64][kCtxCoh];         // region totals: one partial per wave of the reduction
    const int tid = threadIdx.x, c = tid & 7, walker = tid >> 3;
#ifdef DIG_TM_TIMING
    unsigned long long tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm_last = __builtin_readcyclecounter();
#endif
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // Every load of a thread is issued before its first LDS write, and a region's words are requested while the region
    // before it is walked; its description (three loads, then two that depend on the chromosome) is read a region earlier
    // still.  (A load per position inside the code loop was a memory round trip per iteration: 20 us per region and pass, the
    // whole of the first build's 14.7 ms; words fetched when the region is reached: 2 us of every 11.)
    auto request = [&](const CtxRegion& q, uint32_t (&tmp)[kCtxWordsPer]) {
#pragma unroll
        for (int j = 0; j < kCtxWordsPer; ++j) {
            const int64_t i = tid + (int64_t)j * kCtxBlock;
            tmp[j] = i < q.nw ? words[(q.w0 + i < n_words ? q.w0 + i : n_words - 1)] : 0u;
        }
    };
    auto raw_of = [&](int64_t r) {
        CtxRaw a{0, 0, 0};
        if (r < R) {
            a.chrom = reg_chrom[r];
            a.start = reg_start[r];
            a.end = reg_end[r];
        }
        return a;
    };
    const int64_t G = gridDim.x;
    // only_deferred: behind the row walk (dig_tiles_rows.hip), which leaves n_valid = -2 at the regions it did not take: a
    // workgroup visits those of its regions only (and leaves at once when it has none); the mark is replaced in the last pass.
    auto next_region = [&](int64_t r) {
        r += G;
        if (only_deferred)
            while (r < R && n_valid[r] != -2) r += G;
        return r;
    };
    const int64_t r_first = only_deferred ? next_region((int64_t)blockIdx.x - G) : (int64_t)blockIdx.x;
    if (r_first >= R) return;
    for (int64_t c0 = 0; c0 < C; c0 += kCtxCoh) {
        const int cc = (int)(C - c0 < kCtxCoh ? C - c0 : kCtxCoh);
        const bool meta_pass = only_deferred ? c0 + kCtxCoh >= C : c0 == 0;
        __syncthreads();
        for (int idx = tid; idx < (K + 1) * kCtxCoh; idx += kCtxBlock) {
            const int code = idx >> 3, co = idx & 7;
            // code: base d of the window (d = 0 leftmost) at bits 2 d; the reference's index counts the leftmost base highest
            int ref = 0;
#pragma unroll
            for (int d = 0; d < W; ++d) ref = ref * 4 + ((code >> (2 * d)) & 3);
            s_S[code][co] = (code < K && co < cc) ? s_prob[(c0 + co) * K + ref] : 0.0;
        }
        CtxRegion nxt{};
        uint32_t staged[kCtxWordsPer];
        CtxRaw raw1 = raw_of(r_first);
        nxt = ctx_region<U>(raw1, chrom_len[raw1.chrom], chrom_off[raw1.chrom], binsize);
        request(nxt, staged);
        int64_t r1 = next_region(r_first), r2 = next_region(r1);      // the regions after this one
        raw1 = raw_of(r1);
        for (int64_t r = r_first; r < R; r = r1, r1 = r2, r2 = next_region(r2)) {
            const CtxRegion q = nxt;
            const int64_t tiles = q.too_long ? 0 : (q.tiles_valid < n_tiles ? q.tiles_valid : n_tiles);     // tiles written with values
            if (tid == 0 && meta_pass) {
                first_pos[r] = q.first;
                n_valid[r] = q.too_long ? -1 : (int32_t)tiles;
            }
            // (requested now, used after the codes: the description of the region after the next, the chromosome of the next)
            const CtxRaw raw2 = raw_of(r2);
            int64_t len1 = 0, off1 = 0;
            if (r1 < R) {
                len1 = chrom_len[raw1.chrom];
                off1 = chrom_off[raw1.chrom];
            }
            TM_MARK(0);
            __syncthreads();                          // the previous region's walkers are done with s_code / s_sum (and the table is staged)
            TM_MARK(1);
#pragma unroll
            for (int j = 0; j < kCtxWordsPer; ++j) {
                const int64_t i = tid + (int64_t)j * kCtxBlock;
                if (i < q.nw) s_words[i] = staged[j];
            }
            __syncthreads();
            TM_MARK(2);
            // ---- context codes: a thread forms the codes of EIGHT consecutive positions from three words and stores them
            // as one 16-byte piece.  The sixteen nibbles from the first window's leftmost base on are squeezed to sixteen
            // 2-bit fields (five mask-and-fold steps), every (2 W)-bit substring of which IS a code; a window that holds a
            // nibble > 3 gets the zero row (looked at only when the sixteen nibbles hold one). ----
            if (!q.too_long) {
                const int sh = 4 * (int)((q.g0 - U) & 7);        // bit of the first window's leftmost base in s_words[0]
                for (int64_t j = tid; 8 * j < q.n_pos; j += kCtxBlock) {
                    const uint32_t wa = s_words[j], wb = s_words[j + 1], wc = s_words[j + 2];
                    const uint32_t lo = (uint32_t)((((uint64_t)wb << 32) | wa) >> sh), hi = (uint32_t)((((uint64_t)wc << 32) | wb) >> sh);
                    const uint64_t n64 = ((uint64_t)hi << 32) | lo;          // nibbles of the bases 8 j + sh / 4 .. + 15
                    uint64_t y = n64 & 0x3333333333333333ull;
                    y = (y | (y >> 2)) & 0x0f0f0f0f0f0f0f0full;
                    y = (y | (y >> 4)) & 0x00ff00ff00ff00ffull;
                    y = (y | (y >> 8)) & 0x0000ffff0000ffffull;
                    const uint32_t z = (uint32_t)(y | (y >> 16));            // base i at bits 2 i
                    uint32_t code[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) code[k] = (z >> (2 * k)) & (uint32_t)(K - 1);
                    const uint64_t hb = n64 & 0xccccccccccccccccull;
                    if (hb) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if ((hb >> (4 * k)) & ((1ull << (4 * W)) - 1)) code[k] = (uint32_t)K;
                    }
                    *reinterpret_cast<uint4*>(s_code + 8 * j) = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16),
                                                                           code[4] | (code[5] << 16), code[6] | (code[7] << 16));
                }
            }
            __syncthreads();
            TM_MARK(3);
            if (r1 < R) {                             // the next region's words travel during this region's walk
                nxt = ctx_region<U>(raw1, len1, off1, binsize);
                request(nxt, staged);
            }
            raw1 = raw2;
            // the sum of the table values of positions [p0, p1) for cohort lane c
            auto span_sum = [&](int64_t p0, int64_t p1) {
                double acc = 0.0;
                int64_t p = p0;
                // sixteen positions per trip: their codes, then their table rows, are all in flight together (two dependent LDS
                // round trips per trip instead of per position: with two waves per SIMD nothing else hides them); the rest of a
                // span in one trip too: positions past its end read the zero row
                while (p1 - p >= 4) {
                    unsigned k[16];
                    double v[16];
                    if (!(p & 1)) {                   // codes in pairs
                        const uint32_t* c32 = reinterpret_cast<const uint32_t*>(s_code + p);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint32_t two = c32[i];
                            k[2 * i] = two & 0xffffu;
                            k[2 * i + 1] = two >> 16;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) k[i] = s_code[p + i];
                    }
                    if (p1 - p < 16) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) k[i] = p + i < p1 ? k[i] : (unsigned)K;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = s_S[k[i]][c];
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc += v[i];        // (+ 0.0 leaves the sum as it is)
                    p = p + 16 < p1 ? p + 16 : p1;
                }
                if (p < p1) {                         // at most three positions left: codes, then rows, together as well
                    unsigned k[3];
                    double v[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) k[i] = p + i < p1 ? s_code[p + i] : (unsigned)K;
#pragma unroll
                    for (int i = 0; i < 3; ++i) v[i] = s_S[k[i]][c];
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc += v[i];
                }
                return acc;
            };
            // The walkers take the tiles in rounds of kCtxWalkers.  The tiles of a last, short round are cut into pieces so
            // that every walker of that round has one -- a piece of seven positions is one trip where a tile of fifty is four: the
            // busiest walker then makes 13 trips instead of 16.  A split tile's value is the sum of its pieces in order.
            const int64_t whole = q.tiles_valid & ~(int64_t)(kCtxWalkers - 1), rest = q.tiles_valid - whole;
            int pieces = 1;
            if (rest > 0 && rest <= kCtxWalkers / 2 && binsize >= 8 && q.tiles_valid <= kCtxSumTiles) {
                pieces = 2;
                while (pieces < 8 && pieces * 2 * rest <= kCtxWalkers && pieces * 2 * 4 <= binsize) pieces *= 2;
            }
            const int64_t piece_len = (binsize + pieces - 1) / pieces;
            const bool stash = q.tiles_valid <= kCtxSumTiles;
            // ---- region totals over ALL tiles (also those beyond n_tiles, if the caller asked for fewer) ----
            double mine = 0.0;
            if (!q.too_long) {
                const int64_t full_tiles = pieces > 1 ? whole : q.tiles_valid;
                for (int64_t t = walker; t < full_tiles; t += kCtxBlock / 8) {
                    const int64_t p0 = t * binsize, p1 = p0 + binsize < q.n_pos ? p0 + binsize : q.n_pos;
                    const double a = span_sum(p0, p1);
                    mine += a;
                    if (stash) s_sum[c][t] = a;
                }
                if (pieces > 1 && walker < rest * pieces) {
                    const int64_t t = whole + walker / pieces, j = walker % pieces;
                    const int64_t t1 = (t + 1) * binsize < q.n_pos ? (t + 1) * binsize : q.n_pos;
                    int64_t p0 = t * binsize + j * piece_len, p1 = p0 + piece_len;
                    if (p1 > t1) p1 = t1;
                    const double a = p0 < p1 ? span_sum(p0, p1) : 0.0;
                    mine += a;
                    s_sum[c][kCtxSumStride + walker] = a;
                }
            }
            s_part[walker][c] = mine;
            TM_MARK(4);
            __syncthreads();
            TM_MARK(5);
            if (tid < kCtxWalkers) {                  // lane (g, c): walkers 8 g .. 8 g + 7, then the eight g's of the wave
                const int g = tid >> 3;
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += s_part[8 * g + w][c];
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if ((tid & 63) < kCtxCoh) s_T[tid >> 6][tid & 63] = v;
            }
            __syncthreads();
            TM_MARK(6);
            // ---- pt = sum / total: wave w writes cohort w's plane, a lane = a tile (with a thread = a tile for all eight cohorts
            // 200 of the 512 threads wrote and the rest waited at the next barrier) ----
            if (stash || q.too_long) {
                const int co = (tid >> 6) & 7;        // (waves 8 .. 15: the second 64 tiles of every 128)
                if (co < cc) {
                    double total = s_T[0][co];
#pragma unroll
                    for (int w = 1; w < kCtxWalkers / 64; ++w) total += s_T[w][co];
                    double* plane = pt + ((c0 + co) * R + r) * n_tiles;
                    for (int64_t t = (tid & 63) + 64 * (tid >> 9); t < n_tiles; t += kCtxBlock / 8) {
                        double v = nan;
                        if (t < tiles) {
                            if (pieces > 1 && t >= whole) {
                                // (all reads first: one after the other they were 64 LDS round trips for the threads of the
                                //  split tiles, and the whole workgroup waited for them at the next barrier)
                                double pc[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) pc[j] = j < pieces ? s_sum[co][kCtxSumStride + (t - whole) * pieces + j] : 0.0;
                                v = pc[0];
#pragma unroll
                                for (int j = 1; j < 8; ++j) v += pc[j];          // (+ 0.0 leaves the sum as it is)
                            } else {
                                v = s_sum[co][t];
                            }
                            v = v / total;
                        }
                        plane[t] = v;
                    }
                }
            } else {
                for (int64_t tb = 0; tb < n_tiles; tb += kCtxSumTiles) {
                    __syncthreads();
                    const int64_t te = tb + kCtxSumTiles < tiles ? tb + kCtxSumTiles : tiles;
                    for (int64_t t = tb + walker; t < te; t += kCtxBlock / 8) {
                        const int64_t p0 = t * binsize, p1 = p0 + binsize < q.n_pos ? p0 + binsize : q.n_pos;
                        s_sum[c][t - tb] = span_sum(p0, p1);
                    }
                    __syncthreads();
                    const int64_t tn = tb + kCtxSumTiles < n_tiles ? tb + kCtxSumTiles : n_tiles;
                    for (int64_t t = tb + tid; t < tn; t += kCtxBlock)
#pragma unroll
                        for (int co = 0; co < kCtxCoh; ++co)
                            if (co < cc) {
                                double total = s_T[0][co];
#pragma unroll
                                for (int w = 1; w < kCtxWalkers / 64; ++w) total += s_T[w][co];
                                pt[((c0 + co) * R + r) * n_tiles + t] = t < tiles ? s_sum[co][t - tb] / total : nan;
                            }
                }
            }
        }
    }
#ifdef DIG_TM_TIMING
    if (tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_tm_prof[k], tm_acc[k]);
#endif
}

__global__ __launch_bounds__(256) void tile_mut_counts_kernel(const int32_t* __restrict__ pair_mut, const int32_t* __restrict__ pair_reg,
                                                              int64_t n_pairs, const int64_t* __restrict__ mut_start,
                                                              const int32_t* __restrict__ mut_cohort, const int64_t* __restrict__ first_pos,
                                                              const int32_t* __restrict__ n_valid, int binsize, int64_t n_tiles, int64_t R,
                                                              int64_t C, int32_t* __restrict__ k)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += stride) {
        const int64_t m = pair_mut[i], r = pair_reg[i];
        const int64_t off = mut_start[m] - first_pos[r];           // START must be one of the region's positions (:160-163)
        if (off < 0) continue;
        const int64_t t = off / binsize;
        if (t >= n_valid[r]) continue;
        const int64_t c = mut_cohort[m];
        if ((uint64_t)c >= (uint64_t)C) continue;                  // a cohort id outside [0, C) has no plane to count in
        atomicAdd(&k[(c * R + r) * n_tiles + t], 1);
    }
}

}  // namespace dig

using namespace dig;

extern "C" {

#ifdef DIG_TM_TIMING
int dig_debug_tile_profile(unsigned long long* out8)
{
    DIG_HIP_TRY(hipDeviceSynchronize());
    DIG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tm_prof), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DIG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tm_prof), z, sizeof(z)));
    return DIG_OK;
}
#endif

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
    static const bool classic = []() {
        const char* e = getenv("DIG_TILES_FORM");        // developer switch: "classic" forces the vector-FMA kernel
        return e && e[0] == 'c';
    }();
    const bool mfma = !classic && binsize >= 2 && n_tiles >= 1 && n_tiles <= kTileBlock && C >= 1 &&
                      n_tiles * binsize <= (int64_t)(kTileMaxWords - 1) * 8;
    if (mfma) {
        // Which matrix kernel: the two-role one for a chunk of two or more cohort tiles (32 cohorts and up: 1.28 -> 1.22 ms at
        // 37, 1.41 -> 1.34 at 48), the one-role one below that (its two workgroups per CU walk twice as fast when the product
        // is short: 0.93 against 0.95 ms at 21 cohorts, 0.60 against 0.69 at 5).  DIG_TILES_FORM = "one-role" / "two-role"
        // (developer switch) forces either.
        static const int forced = []() {
            const char* e = getenv("DIG_TILES_FORM");
            return e && e[0] == 'o' ? 1 : (e && e[0] == 't' ? 2 : 0);
        }();
        for (int64_t c0 = 0; c0 < C; c0 += kTmChunk) {
            // the chunk's cohorts as full tiles + quads: a remainder of 1 .. 4 is one quad, 5 .. 8 two, 9 and more a (padded) tile.
            // DIG_TILES_CUT=0 (developer switch): whole tiles only.
            static const bool cut = !(getenv("DIG_TILES_CUT") && getenv("DIG_TILES_CUT")[0] == '0');
            const int rem = (int)(C - c0 < kTmChunk ? C - c0 : kTmChunk);
            int mt = rem >> 4, nq = 0;
            const int r16 = rem & 15;
            if (!cut || r16 >= 9) mt += r16 > 0;
            else nq = (r16 + 3) >> 2;
            const bool one_role = forced ? forced == 1 : mt < 2;
            const int grid = one_role ? grid_for(R * kTileBlock, kTileBlock, DIG_TM_OCC) : grid_for(R * kTsBlock, kTsBlock, 1);
            auto go = [&](auto kern, auto kern_roles) {
                if (one_role)
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTileBlock), 0, (hipStream_t)stream, genome_words, n_words, chrom_off,
                                       chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, (int)c0, binsize, (int)n_tiles, pt,
                                       first_pos, n_valid);
                else
                    hipLaunchKernelGGL(kern_roles, dim3(grid), dim3(kTsBlock), 0, (hipStream_t)stream, genome_words, n_words,
                                       chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, (int)c0, binsize,
                                       (int)n_tiles, pt, first_pos, n_valid);
            };
            auto pick = [&](auto mt_c) {
                constexpr int M = decltype(mt_c)::value;
                if (nq == 2) go(base_tile_probs_mfma_kernel<M, 2>, base_tile_probs_roles_kernel<M, 2>);
                else if (nq == 1) go(base_tile_probs_mfma_kernel<M, 1>, base_tile_probs_roles_kernel<M, 1>);
                else if constexpr (M > 0) go(base_tile_probs_mfma_kernel<M, 0>, base_tile_probs_roles_kernel<M, 0>);
            };
            if (mt == 3) go(base_tile_probs_mfma_kernel<3, 0>, base_tile_probs_roles_kernel<3, 0>);
            else if (mt == 2) pick(std::integral_constant<int, 2>{});
            else if (mt == 1) pick(std::integral_constant<int, 1>{});
            else pick(std::integral_constant<int, 0>{});
        }
        DIG_HIP_TRY(hipGetLastError());
        return DIG_OK;
    }
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

int dig_base_tile_probs_ctx(const uint32_t* genome_words, int64_t n_words, const int64_t* chrom_off, const int64_t* chrom_len,
                            int n_chrom, const int32_t* reg_chrom, const int64_t* reg_start, const int64_t* reg_end, int64_t R,
                            const double* s_prob, int64_t C, int n_up, int binsize, int64_t n_tiles, double* pt, int64_t* first_pos,
                            int32_t* n_valid, void* stream)
{
    DIG_REQUIRE(n_up == 1 || n_up == 2, "n_up = n_down = 1 (trinucleotide) or 2 (penta-nucleotide)");
    // developer switches: DIG_TILES_FORM = "general": the general kernel for every region, n_up = 1 too; "rows": the row walk for
    // n_up = 1 too (both cross-check the trinucleotide kernels)
    static const int form = []() {
        const char* e = getenv("DIG_TILES_FORM");
        return e && e[0] == 'g' ? 1 : (e && e[0] == 'r' ? 2 : 0);
    }();
    if (n_up == 1 && form == 0)
        return dig_base_tile_probs(genome_words, n_words, chrom_off, chrom_len, n_chrom, reg_chrom, reg_start, reg_end, R, s_prob, C,
                                   binsize, n_tiles, pt, first_pos, n_valid, stream);
    DIG_REQUIRE(R >= 0 && C >= 0 && n_words >= 2 && n_chrom >= 0 && n_tiles >= 0, "non-negative sizes, n_words >= 2 (pad words)");
    DIG_REQUIRE(binsize >= 1, "binsize >= 1");
    if (R == 0) return DIG_OK;
    DIG_REQUIRE(genome_words && chrom_off && chrom_len && reg_chrom && reg_start && reg_end && first_pos && n_valid, "non-null pointers");
    DIG_REQUIRE(C == 0 || n_tiles == 0 || (s_prob && pt), "s_prob and pt");
    // The row walk (dig_tiles_rows.hip) takes every region its LDS budget covers and marks the others n_valid = -2; the general
    // kernel behind it takes those (a region of more than kCtxMaxPos positions is not evaluated -- n_valid -1, pt NaN; the host
    // wrappers, which know the coordinates, refuse such regions before the launch).
    const int only_deferred = form == 1 ? 0 : 1;
    if (only_deferred) {
        const int rc = launch_tile_probs_rows(genome_words, n_words, chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, n_up,
                                              binsize, n_tiles, pt, first_pos, n_valid, (hipStream_t)stream);
        if (rc != DIG_OK) return rc;
    }
    const int grid = grid_for(R * kCtxBlock, kCtxBlock, 1);
    if (n_up == 1)
        hipLaunchKernelGGL((base_tile_probs_ctx_kernel<1>), dim3(grid), dim3(kCtxBlock), 0, (hipStream_t)stream, genome_words, n_words,
                           chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, binsize, n_tiles, pt, first_pos, n_valid,
                           only_deferred);
    else
        hipLaunchKernelGGL((base_tile_probs_ctx_kernel<2>), dim3(grid), dim3(kCtxBlock), 0, (hipStream_t)stream, genome_words, n_words,
                           chrom_off, chrom_len, reg_chrom, reg_start, reg_end, R, s_prob, C, binsize, n_tiles, pt, first_pos, n_valid,
                           only_deferred);
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
                       n_pairs, mut_start, mut_cohort, first_pos, n_valid, binsize, n_tiles, R, C, k);
    DIG_HIP_TRY(hipGetLastError());
    return DIG_OK;
}

}  // extern "C"
